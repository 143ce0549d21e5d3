"""Marching-cubes case table, DERIVED (not transcribed): the generator of ucnerf_amd/csrc/mc_table.h
(`python tools/gen_mc_table.py` rewrites the header; tests/test_mesh.py checks that the committed header is this output, and
the host restatement used by the tests takes the same table from here).

The reference takes its meshes from skimage.measure.marching_cubes (extract.py:379-383, tsdf.py:98-102: Lewiner's variant, a
third-party library that is not installed here), so there is no table of the reference's own to follow.  This one is built from
first principles so that it is crack-free by construction:

  * corner c of a cell sits at (c & 1, (c >> 1) & 1, (c >> 2) & 1); edge e = 4 a + k runs along axis a from the corner whose two
    other coordinates are the bits of k;
  * a case is the 8-bit set of corners INSIDE the surface (value < level);
  * on every face the cut edges are joined pairwise; a face with four cuts (two diagonal corners inside) is ambiguous and is
    resolved by a rule that looks at that face's four corner signs only -- each inside corner is cut off on its own -- so the
    two cells sharing the face always agree and the surface has no holes (the 1987 table does);
  * the segments close into loops (every cut edge lies on exactly two faces); each loop is oriented so that its normal points
    from the inside corners to the outside ones, and triangulated as a fan.
"""
import itertools
import os

import numpy as np

CORNERS = [(c & 1, (c >> 1) & 1, (c >> 2) & 1) for c in range(8)]


def corner_id(p):
    return p[0] | (p[1] << 1) | (p[2] << 2)


def edge_corners(e):
    a, k = divmod(e, 4)
    others = [d for d in range(3) if d != a]
    p0 = [0, 0, 0]
    p0[others[0]] = k & 1
    p0[others[1]] = (k >> 1) & 1
    p1 = list(p0)
    p1[a] = 1
    return corner_id(p0), corner_id(p1)


EDGES = [edge_corners(e) for e in range(12)]
EDGE_OF = {frozenset(ec): e for e, ec in enumerate(EDGES)}


def faces():
    """Each face as its 4 corners in cyclic order."""
    out = []
    for a in range(3):
        u, v = [d for d in range(3) if d != a]
        for side in (0, 1):
            ring = []
            for (du, dv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[a], p[u], p[v] = side, du, dv
                ring.append(corner_id(p))
            out.append(ring)
    return out


FACES = faces()


def case_triangles(case):
    inside = [(case >> c) & 1 for c in range(8)]
    seg = {}                                             # edge -> its (up to two) neighbours along the surface

    def link(e0, e1):
        seg.setdefault(e0, []).append(e1)
        seg.setdefault(e1, []).append(e0)
    for ring in FACES:
        cuts = []                                        # (position in the ring, edge) of the ring edges that are cut
        for i in range(4):
            c0, c1 = ring[i], ring[(i + 1) % 4]
            if inside[c0] != inside[c1]:
                cuts.append((i, EDGE_OF[frozenset((c0, c1))]))
        if len(cuts) == 2:
            link(cuts[0][1], cuts[1][1])
        elif len(cuts) == 4:
            # ring edges i-1 and i meet at corner ring[i]: cut off every INSIDE corner on its own
            for i in range(4):
                if inside[ring[i]]:
                    link(EDGE_OF[frozenset((ring[(i - 1) % 4], ring[i]))], EDGE_OF[frozenset((ring[i], ring[(i + 1) % 4]))])
    assert all(len(v) == 2 for v in seg.values()), (case, seg)
    tris, seen = [], set()
    mid = lambda e: (np.array(CORNERS[EDGES[e][0]], float) + np.array(CORNERS[EDGES[e][1]], float)) / 2
    for start in sorted(seg):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [n for n in seg[cur] if n != prev]
            nxt = nxt[0] if nxt else seg[cur][0]
            if len(seg[cur]) == 2 and seg[cur][0] == seg[cur][1]:
                nxt = seg[cur][0]
            if nxt == start:
                break
            loop.append(nxt)
            seen.add(nxt)
            prev, cur = cur, nxt
        assert len(loop) >= 3, (case, loop)
        # orientation: the loop's normal (Newell) must point from the inside end of its edges to the outside end
        pts = [mid(e) for e in loop]
        nrm = np.zeros(3)
        for i in range(len(pts)):
            a, b = pts[i], pts[(i + 1) % len(pts)]
            nrm += np.cross(a, b)
        grad = np.zeros(3)
        for e in loop:
            c0, c1 = EDGES[e]
            cin, cout = (c0, c1) if inside[c0] else (c1, c0)
            grad += np.array(CORNERS[cout], float) - np.array(CORNERS[cin], float)
        assert abs(float(nrm @ grad)) > 1e-9, (case, loop)
        if nrm @ grad < 0:
            loop = loop[::-1]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    return tris


def build():
    """(tri_table int8 [256, 16] = up to 5 triangles as edge triples, -1 terminated; n_tri uint8 [256]; edge_corners [12, 2])."""
    table = -np.ones((256, 16), np.int8)
    count = np.zeros(256, np.uint8)
    for case in range(256):
        tris = case_triangles(case)
        assert len(tris) <= 5, (case, len(tris))
        count[case] = len(tris)
        for i, t in enumerate(tris):
            table[case, 3 * i:3 * i + 3] = t
    return table, count, np.array(EDGES, np.uint8)


def header_text():
    table, count, edges = build()
    lines = ["// GENERATED by tools/gen_mc_table.py (python tools/gen_mc_table.py) -- do not edit.  Marching-cubes case table derived from",
             "// first principles (face-consistent resolution of the ambiguous faces: crack-free); see that file for the conventions.",
             "#pragma once", "#include <stdint.h>", "#ifndef MC_TABLE_QUALIFIER", "#define MC_TABLE_QUALIFIER static const", "#endif", "",
             "// corner c at (c & 1, (c >> 1) & 1, (c >> 2) & 1); edge e = 4 axis + k joins these two corners",
             "MC_TABLE_QUALIFIER uint8_t kMcEdgeCorners[12][2] = {" + ", ".join("{%d, %d}" % tuple(e) for e in edges) + "};",
             "MC_TABLE_QUALIFIER uint8_t kMcTriCount[256] = {" + ", ".join(str(int(c)) for c in count) + "};",
             "MC_TABLE_QUALIFIER int8_t kMcTriTable[256][16] = {"]
    for case in range(256):
        lines.append("    {" + ", ".join("%2d" % int(v) for v in table[case]) + "},")
    lines.append("};")
    return "\n".join(lines) + "\n"


HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ucnerf_amd", "csrc", "mc_table.h")

if __name__ == "__main__":
    open(HEADER, "w").write(header_text())
    t, c, _ = build()
    print("wrote", HEADER, "max triangles per case", int(c.max()), "total", int(c.sum()))
