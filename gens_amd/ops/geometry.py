"""K11 (lattice points) and K12 (iso-surface extraction on the device).

Part of gens_amd.ops (see ops/__init__.py); citations are relative to /root/reference."""
from .base import *  # noqa: F401,F403

# ------------------------------------------------------------------------------------------------------------------
# K11  lattice (implicit_surface.py:407-418)
# ------------------------------------------------------------------------------------------------------------------
def lattice_points(bound_min, bound_max, resolution, first, count, device):
    lo = (C.c_float * 3)(*[float(v) for v in bound_min])
    hi = (C.c_float * 3)(*[float(v) for v in bound_max])
    pts = torch.empty(count, 3, device=device, dtype=_f32)
    L.call("gens_lattice_points", lo, hi, int(resolution), int(first), int(count), L.ptr(pts), L.stream())
    return pts


# ------------------------------------------------------------------------------------------------------------------
# K12  iso-surface extraction (mcubes.marching_cubes at implicit_surface.py:423)
# ------------------------------------------------------------------------------------------------------------------
_MC_TABLES = {}


def marching_cubes(u, threshold=0.0):
    """u (X,Y,Z) float32 device tensor -> (vertices (V,3) float64 in index coordinates, triangles (T,3) int32), both on the
    device.  Classic marching cubes with the case table of gens_amd/mc_tables.py; order as in oracle/mc_oracle.py."""
    from .. import mc_tables
    dev = u.device
    if dev not in _MC_TABLES:
        _MC_TABLES[dev] = (torch.from_numpy(mc_tables.TRI_TABLE.copy()).to(dev), torch.from_numpy(mc_tables.TRI_COUNT.copy()).to(dev))
    table, count = _MC_TABLES[dev]
    u = _c(u.detach().to(_f32))
    x, y, z = u.shape
    total = x * y * z
    vmask, vcount, cases, tcount = (torch.empty(total, device=dev, dtype=torch.uint8) for _ in range(4))
    u8 = torch.uint8
    L.call("gens_mc_classify", L.ptr(u), x, y, z, float(threshold), L.ptr(count, u8), L.ptr(vmask, u8), L.ptr(vcount, u8), L.ptr(cases, u8),
           L.ptr(tcount, u8), L.stream(), nbytes=total * 8)
    vend = torch.cumsum(vcount, 0, dtype=torch.int32)
    tend = torch.cumsum(tcount, 0, dtype=torch.int32)
    nv, nt = int(vend[-1]), int(tend[-1])
    vertices = torch.empty(nv, 3, device=dev, dtype=torch.float64)
    triangles = torch.empty(nt, 3, device=dev, dtype=torch.int32)
    if nv == 0:
        return vertices, triangles
    voff = vend - vcount        # exclusive scans
    toff = tend - tcount
    del vend, tend
    i32 = torch.int32
    L.call("gens_mc_emit", L.ptr(u), x, y, z, float(threshold), L.ptr(table, torch.int8), table.shape[1], L.ptr(vmask, u8), L.ptr(voff, i32),
           L.ptr(cases, u8), L.ptr(tcount, u8), L.ptr(toff, i32), L.ptr(vertices, torch.float64),
           L.ptr(triangles, i32) if nt else L.ptr(torch.empty(1, 3, device=dev, dtype=i32), i32), L.stream(),
           nbytes=total * 15 + nv * 24 + nt * 12)
    return vertices, triangles


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]      # private helpers travel too: the package namespace is the old module's
