#!/usr/bin/env python3
"""Generate the marching-cubes case table used by csrc/tsdf.hip and oracle/tsdf.py (-> gaussiananything_amd/csrc/mc_table.h,
oracle/mc_table.json).

The reference extracts its mesh with Open3D's ScalableTSDFVolume.extract_triangle_mesh (third party, absent here; its case
table is the classic 256 x 16 one, which cannot be reproduced from memory).  The table is therefore DERIVED, by rule:

* corner i of a cube sits at (i & 1, (i >> 1) & 1, (i >> 2) & 1); it is "inside" when its tsdf is negative (bit i of the case);
* edge e = 4 * axis + 2 * b + a joins the corners that differ along `axis`, the other two coordinates (in x, y, z order with
  `axis` removed) being a and b;
* on every face the intersected edges are joined in pairs; a face with two diagonal inside corners (the ambiguous case) cuts
  the inside corners off separately -- a rule that depends only on the face's four signs, so neighbouring cubes agree and the
  surface is closed;
* the face segments chain into closed loops; every loop is triangulated -- as a fan from its smallest edge id unless that lays a
  diagonal or a whole triangle into a cube face, in which case the triangulation with the fewest such diagonals is taken --
  and oriented so that the normal points from the inside (negative) to the outside (positive) corners.

The vertices of the mesh (zero crossings on intersected edges of valid cubes) do not depend on this table; only the choice
of diagonals inside a cube does."""
import itertools
import json
import os

import numpy as np

CORNER = np.array([[i & 1, (i >> 1) & 1, (i >> 2) & 1] for i in range(8)])


def edge_id(c0, c1):
    d = CORNER[c0] ^ CORNER[c1]
    axis = int(np.argmax(d))
    other = [k for k in range(3) if k != axis]
    a, b = CORNER[c0][other[0]], CORNER[c0][other[1]]
    return 4 * axis + 2 * int(b) + int(a)


EDGE_CORNERS = {}
for c0, c1 in itertools.combinations(range(8), 2):
    if int((CORNER[c0] ^ CORNER[c1]).sum()) == 1:
        EDGE_CORNERS[edge_id(c0, c1)] = (c0, c1)


def faces():
    out = []
    for axis in range(3):
        other = [k for k in range(3) if k != axis]
        for side in (0, 1):
            cyc = []
            for a, b in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[axis] = side
                p[other[0]] = a
                p[other[1]] = b
                cyc.append(p[0] | (p[1] << 1) | (p[2] << 2))
            out.append(cyc)
    return out


FACES = faces()


def edge_faces(e):
    """the (axis, side) cube faces that contain edge e"""
    c0, c1 = EDGE_CORNERS[e]
    return {(a, int(CORNER[c0][a])) for a in range(3) if CORNER[c0][a] == CORNER[c1][a]}


def _triangulations(idx):
    """all triangulations of the polygon idx[0..n-1] (index tuples)"""
    if len(idx) < 3:
        return [[]]
    if len(idx) == 3:
        return [[tuple(idx)]]
    out = []
    for k in range(1, len(idx) - 1):      # the triangle on edge (idx[0], idx[-1]) has apex idx[k]
        for left in _triangulations(idx[:k + 1]):
            for right in _triangulations(idx[k:]):
                out.append(left + [(idx[0], idx[k], idx[-1])] + right)
    return out


def best_triangulation(loop):
    """Triangulate the loop (edge ids in cyclic order) without laying triangle edges -- let alone whole triangles -- into a
    cube face where that can be avoided: a diagonal inside a face is matched by nothing in the neighbouring cube (or by the
    neighbour's own such diagonal: an edge with four triangles).  First choice among equals: the fan from loop[0]."""
    n = len(loop)
    poly_edges = {frozenset((loop[i], loop[(i + 1) % n])) for i in range(n)}

    def cost(tris):
        c = 0
        for t in tris:
            if len(set.intersection(*(edge_faces(loop[i]) for i in t))) > 0:
                c += 100                                           # a triangle lying in a cube face
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                if frozenset((loop[a], loop[b])) not in poly_edges and edge_faces(loop[a]) & edge_faces(loop[b]):
                    c += 1                                         # (counted from both sides) a diagonal inside a cube face
        return c

    fan = [(0, k, k + 1) for k in range(1, n - 1)]
    best, best_cost = fan, cost(fan)
    if best_cost:
        for tris in _triangulations(list(range(n))):
            cc = cost(tris)
            if cc < best_cost:
                best, best_cost = tris, cc
    return [(loop[a], loop[b], loop[c]) for a, b, c in best]


def case_triangles(case):
    inside = [(case >> i) & 1 for i in range(8)]
    adj = {}

    def link(e0, e1):
        adj.setdefault(e0, []).append(e1)
        adj.setdefault(e1, []).append(e0)

    for cyc in FACES:
        cut = [(k, edge_id(cyc[k], cyc[(k + 1) % 4])) for k in range(4) if inside[cyc[k]] != inside[cyc[(k + 1) % 4]]]
        if len(cut) == 2:
            link(cut[0][1], cut[1][1])
        elif len(cut) == 4:
            # ambiguous face: each inside corner is cut off by the two edges that meet in it
            for k in range(4):
                if inside[cyc[k]]:
                    link(edge_id(cyc[(k - 1) % 4], cyc[k]), edge_id(cyc[k], cyc[(k + 1) % 4]))
    tris = []
    seen = set()
    for start in sorted(adj):
        if start in seen:
            continue
        loop = [start]
        seen.add(start)
        prev, cur = None, start
        while True:
            assert len(adj[cur]) == 2 and adj[cur][0] != adj[cur][1], (case, cur, adj[cur])
            step = adj[cur][0] if adj[cur][0] != prev else adj[cur][1]
            if step == start:
                break
            loop.append(step)
            seen.add(step)
            prev, cur = cur, step
        pts = {e: (CORNER[EDGE_CORNERS[e][0]] + CORNER[EDGE_CORNERS[e][1]]) / 2.0 for e in loop}
        fan = best_triangulation(loop)
        score = 0.0
        for a, b, c in fan:
            n = np.cross(pts[b] - pts[a], pts[c] - pts[a])
            for e in (a, b, c):
                c0, c1 = EDGE_CORNERS[e]
                d = (CORNER[c1] - CORNER[c0]) * (1.0 if inside[c0] else -1.0)   # inside -> outside
                score += float(n @ d)
        if score < 0:
            fan = [(a, c, b) for a, b, c in fan]
        tris.extend(fan)
    return tris


def main():
    table = [case_triangles(c) for c in range(256)]
    ntri = max(len(t) for t in table)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = []
    for t in table:
        flat = [e for tri in t for e in tri]
        rows.append(flat + [-1] * (3 * ntri - len(flat)))
    with open(os.path.join(root, "oracle", "mc_table.json"), "w") as f:
        json.dump({"max_triangles": ntri, "edge_corners": [list(EDGE_CORNERS[e]) for e in range(12)], "triangles": rows}, f)
    with open(os.path.join(root, "gaussiananything_amd", "csrc", "mc_table.h"), "w") as f:
        f.write("// GENERATED by tools/gen_mc_table.py (the rule is stated there) -- do not edit.\n#pragma once\n#include <cstdint>\n")
        f.write(f"namespace ga {{\nconstexpr int kMcMaxTriangles = {ntri};\n")
        f.write("// corners joined by edge e = 4 axis + 2 b + a (corner i at (i & 1, (i >> 1) & 1, (i >> 2) & 1))\n")
        f.write("__device__ __constant__ const int8_t kMcEdgeCorners[12][2] = {" + ", ".join("{%d, %d}" % EDGE_CORNERS[e] for e in range(12)) + "};\n")
        f.write("__device__ __constant__ const int8_t kMcTriangleCount[256] = {" + ", ".join(str(len(t)) for t in table) + "};\n")
        f.write(f"__device__ __constant__ const int8_t kMcTriangles[256][{3 * ntri}] = {{\n")
        for r in rows:
            f.write("    {" + ", ".join(str(v) for v in r) + "},\n")
        f.write("};\n}  // namespace ga\n")
    print("max triangles per cube", ntri, "total", sum(len(t) for t in table))


if __name__ == "__main__":
    main()
