"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the iso-surface extraction behind
`ImplicitSurface.extract_geometry` (/root/reference/models/modules/implicit_surface.py:407-427): the reference calls
`mcubes.marching_cubes(u, threshold)` (PyMCubes==0.1.4, requirements.txt:11; third-party, absent from /root/reference
and from this image, so this part of the oracle is "parity unpinned": there is no golden mesh to compare with).

What is restated is the published Lorensen-Cline algorithm as PyMCubes implements it:
  * case index bit n set iff u[corner n] < iso (Bourke corner numbering);
  * one vertex per lattice edge whose end values straddle iso, at  a + (iso - u_a) / (u_b - u_a)  along the edge,
    evaluated in float64 from the float32 samples (PyMCubes interpolates in double), in INDEX coordinates;
  * triangles of a cell from a 256-entry case table (passed in; the product generates it in gens_amd/mc_tables.py).
Ordering (defined here, the same in the HIP kernels): vertices by (owner lattice point in C order, axis x<y<z);
triangles by (cell in C order, table order).  Only tests/ may import this module.
"""
import numpy as np

_EDGE_OWNER = np.array([(0, 0, 0, 0), (1, 0, 0, 1), (0, 1, 0, 0), (0, 0, 0, 1), (0, 0, 1, 0), (1, 0, 1, 1), (0, 1, 1, 0), (0, 0, 1, 1),
                        (0, 0, 0, 2), (1, 0, 0, 2), (1, 1, 0, 2), (0, 1, 0, 2)], dtype=np.int64)
_CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=np.int64)


def marching_cubes(u, iso, tri_table, tri_count):
    """u (X,Y,Z) float32 -> vertices (V,3) float64 in index coordinates, triangles (T,3) int64."""
    u = np.asarray(u, dtype=np.float32)
    X, Y, Z = u.shape
    below = u < np.float32(iso)
    # --- vertices: edge (p, axis) crosses if below[p] != below[p + e_axis]
    cross = np.zeros((X, Y, Z, 3), dtype=bool)
    cross[:-1, :, :, 0] = below[:-1] != below[1:]
    cross[:, :-1, :, 1] = below[:, :-1] != below[:, 1:]
    cross[:, :, :-1, 2] = below[:, :, :-1] != below[:, :, 1:]
    flat = cross.reshape(-1)
    vid = np.cumsum(flat, dtype=np.int64) - flat            # exclusive: id of edge (p, axis) in (p, axis) order
    vid = vid.reshape(X, Y, Z, 3)
    p = np.argwhere(cross)                                  # rows (i, j, k, axis) in C order
    a = u[p[:, 0], p[:, 1], p[:, 2]].astype(np.float64)
    q = p[:, :3].copy()
    q[np.arange(len(p)), p[:, 3]] += 1
    b = u[q[:, 0], q[:, 1], q[:, 2]].astype(np.float64)
    t = (np.float64(np.float32(iso)) - a) / (b - a)
    vertices = p[:, :3].astype(np.float64)
    vertices[np.arange(len(p)), p[:, 3]] += t
    # --- triangles
    case = np.zeros((X - 1, Y - 1, Z - 1), dtype=np.int64)
    for n, (dx, dy, dz) in enumerate(_CORNERS):
        case |= below[dx:X - 1 + dx, dy:Y - 1 + dy, dz:Z - 1 + dz].astype(np.int64) << n
    cells = np.argwhere(tri_count[case] > 0)
    tris = []
    for (i, j, k) in cells:
        c = case[i, j, k]
        for tt in range(int(tri_count[c])):
            tri = []
            for e in tri_table[c, 3 * tt:3 * tt + 3]:
                ox, oy, oz, ax = _EDGE_OWNER[e]
                tri.append(vid[i + ox, j + oy, k + oz, ax])
            tris.append(tri)
    triangles = np.array(tris, dtype=np.int64).reshape(-1, 3)
    return vertices, triangles
