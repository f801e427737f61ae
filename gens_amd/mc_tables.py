"""Marching-cubes case table, generated (not transcribed): 256 corner configurations -> triangles over the 12 cube edges.

The reference extracts its mesh with PyMCubes (`mcubes.marching_cubes`, /root/reference/models/modules/implicit_surface.py:423,
pinned `PyMCubes==0.1.4` in requirements.txt:11 -- a third-party dependency that is neither vendored in the reference
nor installed in this image).  PyMCubes implements the classic Lorensen-Cline algorithm: a vertex on every lattice edge
whose end values straddle the iso-value (linear interpolation), triangles per cell from a 256-entry case table.  The
vertex set is fully determined by the field; the case table only decides how the crossing points of one cell are
joined.  This module derives a topologically consistent table from first principles:

  * on each cube face the crossing points are joined pairwise; on an ambiguous face (diagonal corners alike) the two
    segments cut off the corners BELOW the iso-value -- a function of the face's corner signs only, so the two cells
    sharing a face always agree and the mesh is watertight;
  * every segment is directed with the below-iso corners of its face on its left (seen from outside the cube), so the
    segments chain into consistently oriented closed loops whose normals point to the below-iso side (for u = -sdf:
    out of the object); each loop is fan-triangulated.

Ambiguous cells may therefore be triangulated differently from PyMCubes' table (same vertices, same surface up to the
choice inside such a cell).  tests/test_marching_cubes.py checks closedness, orientation and Euler characteristic.

Conventions (Bourke / PyMCubes numbering): corner c = (x, y, z) offsets
    c0 000, c1 100, c2 110, c3 010, c4 001, c5 101, c6 111, c7 011;   bit n of the case index is set iff u[c_n] < iso.
"""
import numpy as np

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=np.int64)
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
# every edge is owned by the lattice point at its lower end: (owner offset, axis)
EDGE_OWNER = np.array([(0, 0, 0, 0), (1, 0, 0, 1), (0, 1, 0, 0), (0, 0, 0, 1), (0, 0, 1, 0), (1, 0, 1, 1), (0, 1, 1, 0), (0, 0, 1, 1),
                       (0, 0, 0, 2), (1, 0, 0, 2), (1, 1, 0, 2), (0, 1, 0, 2)], dtype=np.int32)
# faces as corner cycles; the edge between cycle[i] and cycle[i+1] is looked up in EDGES
FACES = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (3, 2, 6, 7), (0, 3, 7, 4), (1, 2, 6, 5)]
MAX_TRIS = 6


def _edge_id(a, b):
    for e, (p, q) in enumerate(EDGES):
        if (p, q) == (a, b) or (p, q) == (b, a):
            return e
    raise KeyError((a, b))


def _ccw_faces():
    """Face corner cycles, counter-clockwise as seen from outside the cube."""
    out = []
    for cyc in FACES:
        p = CORNERS[list(cyc)].astype(np.float64)
        normal = np.cross(p[1] - p[0], p[2] - p[1])
        if np.dot(normal, p.mean(0) - 0.5) < 0:
            cyc = cyc[::-1]
        out.append(tuple(cyc))
    return out


_CCW = _ccw_faces()
_FACE_EDGES = [{_edge_id(c[i], c[(i + 1) % 4]) for i in range(4)} for c in FACES]


def _same_face(e0, e1):
    return any(e0 in f and e1 in f for f in _FACE_EDGES)


def _case(index):
    below = [(index >> n) & 1 for n in range(8)]
    nxt = {}                                             # directed: crossing edge -> next crossing edge of its loop

    def join(e_from, e_to):
        assert e_from not in nxt
        nxt[e_from] = e_to

    # Every segment is directed so that, seen from outside the cube, the below-iso corners of its face lie on its LEFT:
    # the loops then run counter-clockwise around the below-iso region and, by the right-hand rule, the triangle
    # normals point to the below-iso side.  (Walking a chord of a CCW polygon from edge a to edge b, the corners met
    # counter-clockwise between a and b are on the right.)
    for cyc in _CCW:
        edges = [_edge_id(cyc[i], cyc[(i + 1) % 4]) for i in range(4)]
        cross = [i for i in range(4) if below[cyc[i]] != below[cyc[(i + 1) % 4]]]
        if len(cross) == 2:
            a, b = cross
            if below[cyc[(a + 1) % 4]]:                  # corners a+1..b (on the right of a->b) are below: go b->a
                join(edges[b], edges[a])
            else:
                join(edges[a], edges[b])
        elif len(cross) == 4:                            # ambiguous face: cut off each below-iso corner
            for i in range(4):
                if below[cyc[i]]:
                    join(edges[i], edges[(i - 1) % 4])   # corner cyc[i] lies between edges i-1 and i: keep it on the left
    tris = []
    seen = set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, cur = [], start
        while cur not in seen:
            seen.add(cur)
            loop.append(cur)
            cur = nxt[cur]
        assert cur == start and len(loop) >= 3
        # fan apex: the rotation of the loop with the fewest diagonals running inside a cube face (two crossing points
        # of the same face that the face rule did not join) -- such a diagonal would lie in the plane shared with the
        # neighbouring cell
        best = min(range(len(loop)), key=lambda r: (sum(_same_face(loop[r], loop[(r + i) % len(loop)]) for i in range(2, len(loop) - 1)), r))
        loop = loop[best:] + loop[:best]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    return tris


def build():
    """-> (tri_table int8 (256, 3*MAX_TRIS) padded with -1, tri_count uint8 (256,))."""
    table = -np.ones((256, 3 * MAX_TRIS), dtype=np.int8)
    count = np.zeros(256, dtype=np.uint8)
    for index in range(256):
        tris = _case(index)
        assert len(tris) <= MAX_TRIS, (index, len(tris))
        count[index] = len(tris)
        for t, tri in enumerate(tris):
            table[index, 3 * t:3 * t + 3] = tri
    return table, count


TRI_TABLE, TRI_COUNT = build()
