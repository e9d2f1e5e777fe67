#!/usr/bin/env python3
"""Generate the 256-case marching-cubes triangle table used by csrc/surface.hip -> ishapediting_amd/csrc/mc_table.h.

The reference calls PyMCubes (`mcubes.marching_cubes`, triplane_decoder/visualize.py:71,100), which is not installed
here and whose table is not in the reference tree, so the table is DERIVED, not copied: for every corner configuration
the surface polygons are traced on the cube --
  * a vertex sits on each of the 12 cube edges whose end points straddle the level (exactly the marching-cubes vertex
    set: one per sign-changing grid edge, shared between the cells around the edge);
  * on every face the crossing edges are joined by segments; a face with four crossings (the ambiguous case) is cut so
    that each INSIDE corner is separated -- a rule that depends only on the face's own four corner signs, so the two
    cells sharing a face always agree and the mesh is watertight;
  * the segments close into loops (every crossing edge lies on two faces), each loop is fan-triangulated and oriented so
    that its normal points from the inside (value > level) to the outside.
Corner c = (c & 1, (c >> 1) & 1, (c >> 2) & 1) -> (dx, dy, dz); edge e = 4 * axis + k where k enumerates the four edges
parallel to `axis` by the two other coordinates of their lower corner.
Cases with five triangles exist; none needs more (checked below)."""
import os
import sys

import numpy as np

CORNER = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.float64)


def edge_list():
    """12 edges as (lower corner, upper corner, axis); id = 4*axis + rank among the edges of that axis."""
    edges = []
    for axis in range(3):
        for lo in range(8):
            if not (lo >> axis) & 1:
                edges.append((lo, lo | (1 << axis), axis))
    return edges


EDGES = edge_list()
EDGE_ID = {(a, b): i for i, (a, b, _) in enumerate(EDGES)}
# faces: 4 corners in cyclic order (any winding: loops are oriented afterwards from the geometry)
FACES = []
for axis in range(3):
    for side in (0, 1):
        o = [a for a in range(3) if a != axis]
        cyc = []
        for (u, v) in ((0, 0), (1, 0), (1, 1), (0, 1)):
            c = (side << axis) | (u << o[0]) | (v << o[1])
            cyc.append(c)
        FACES.append(cyc)


def eid(a, b):
    return EDGE_ID[(min(a, b), max(a, b))]


def case_triangles(bits):
    inside = [(bits >> c) & 1 for c in range(8)]
    seg = []
    for cyc in FACES:
        cross = [inside[cyc[i]] != inside[cyc[(i + 1) % 4]] for i in range(4)]      # edge i joins corner i and i+1
        n = sum(cross)
        if n == 2:
            i, j = [k for k in range(4) if cross[k]]
            seg.append((eid(cyc[i], cyc[(i + 1) % 4]), eid(cyc[j], cyc[(j + 1) % 4])))
        elif n == 4:
            # corners alternate; cut off each inside corner: its two incident face edges are joined
            for k in range(4):
                if inside[cyc[k]]:
                    seg.append((eid(cyc[(k - 1) % 4], cyc[k]), eid(cyc[k], cyc[(k + 1) % 4])))
    # link the segments into loops
    adj = {}
    for a, b in seg:
        adj.setdefault(a, []).append(b)
        adj.setdefault(b, []).append(a)
    assert all(len(v) == 2 for v in adj.values()), bits
    loops, seen = [], set()
    for start in sorted(adj):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [x for x in adj[cur] if x != prev]
            # a 2-cycle cannot occur (two faces never share two edges); take the unvisited neighbour
            cand = [x for x in nxt if x not in seen]
            if not cand:
                break
            prev, cur = cur, cand[0]
            loop.append(cur)
            seen.add(cur)
        loops.append(loop)
    tris = []
    mid = lambda e: 0.5 * (CORNER[EDGES[e][0]] + CORNER[EDGES[e][1]])
    for loop in loops:
        assert len(loop) >= 3, (bits, loop)
        pts = [mid(e) for e in loop]
        # Newell normal of the loop vs the inside->outside direction summed over its edges
        nrm = np.zeros(3)
        for i in range(len(pts)):
            p, q = pts[i], pts[(i + 1) % len(pts)]
            nrm += np.cross(p, q)
        out_dir = np.zeros(3)
        for e in loop:
            a, b, _ = EDGES[e]
            out_dir += (CORNER[b] - CORNER[a]) * (1.0 if inside[a] else -1.0)
        if np.dot(nrm, out_dir) < 0:
            loop = loop[::-1]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    return tris


def build():
    table = [case_triangles(b) for b in range(256)]
    assert table[0] == [] and table[255] == []
    mx = max(len(t) for t in table)
    assert mx <= 5, mx
    # every case and its complement use the same crossing edges
    for b in range(256):
        ea = sorted({e for t in table[b] for e in t})
        eb = sorted({e for t in table[255 - b] for e in t})
        assert ea == eb, b
    return table


def emit(path):
    table = build()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/make_mc_table.py -- do not edit.  256-case marching-cubes triangle table (see that script for the\n"
                "// derivation: loop tracing with the 'separate the inside corners' face rule; not a copy of any published table).\n"
                "// Corner c = (c & 1, (c >> 1) & 1, (c >> 2) & 1); edge e: lower corner c_mc_edge_lo[e], axis e / 4.\n#pragma once\n")
        f.write("__constant__ unsigned char c_mc_edge_lo[12] = {" + ", ".join(str(a) for a, _, _ in EDGES) + "};\n")
        f.write("__constant__ unsigned char c_mc_ntri[256] = {" + ", ".join(str(len(t)) for t in table) + "};\n")
        f.write("__constant__ signed char c_mc_tri[256][15] = {\n")
        for t in table:
            flat = [e for tri in t for e in tri]
            flat += [-1] * (15 - len(flat))
            f.write("  {" + ", ".join(str(v) for v in flat) + "},\n")
        f.write("};\n")
    print(f"wrote {path}: max {max(len(t) for t in table)} triangles per cell, {sum(len(t) for t in table)} in all")


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emit(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "ishapediting_amd", "csrc", "mc_table.h"))
