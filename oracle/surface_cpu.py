"""TEST INFRASTRUCTURE ONLY -- CPU statement (torch, fp32) of the surface algorithms that ishapediting_amd/csrc/surface.hip
runs on the device: marching tetrahedra (6 per cell, shared vertices at the linear zero crossing), the marching-cubes
vertex set, Open3D-style simple Laplacian smoothing and the Chamfer distance of meshProcess.py:18-35.

Parity status: **unpinned against the reference's third-party calls** (PyMCubes `marching_cubes`, Open3D
`filter_smooth_simple` / surface sampling are not installed here and their versions are not pinned by the reference,
SURVEY.md 8c).  What IS pinned: analytic properties (vertices of a sphere SDF lie on the sphere, the mesh is closed,
Euler characteristic 2) in tests/test_oracle_golden.py, and the HIP kernels against this file in tests/test_gpu_surface.py.
Only tests/, tools/ and bench.py's checker legs may import this module; nothing under ishapediting_amd/ does.
"""
from __future__ import annotations

import torch


def mc_vertices(volume: torch.Tensor, level: float = 0.0) -> torch.Tensor:
    """The vertex set marching cubes produces: one vertex per grid edge whose end points straddle `level`,
    at the linearly interpolated crossing (grid coordinates, [V,3]).  Triangulation tables are not needed for
    vertex counts or for the Chamfer metric of meshProcess.py:18-35.

    Tie rule (a voxel exactly AT the level): a corner is classified by `value <= level` vs `value > level`, the partition
    PyMCubes' published marching_cubes uses (`if (v[m] <= isovalue) cubeindex |= 1 << m;`, _mcubes/marchingcubes.h --
    restated from the published source, PyMCubes is not installed here: parity unpinned) and the one
    csrc/surface.hip:corner_bits and `marching_cubes` below use (`value - level > 0` = the other side of the same cut).
    An at-level voxel therefore sits with the values below the level, and an edge from it to a value above the level
    carries a vertex AT the voxel (t = 0)."""
    v = volume.float() - level
    out = []
    for axis in range(3):
        a = v.narrow(axis, 0, v.shape[axis] - 1)
        b = v.narrow(axis, 1, v.shape[axis] - 1)
        cross = (a > 0) != (b > 0)
        idx = cross.nonzero()
        if idx.numel() == 0:
            continue
        va, vb = a[cross], b[cross]
        t = va / (va - vb)
        p = idx.float()
        p[:, axis] += t
        out.append(p)
    return torch.cat(out, dim=0) if out else torch.zeros((0, 3), device=volume.device)


def chamfer_distance(pa: torch.Tensor, pb: torch.Tensor, point_num=20000, seed: int = 0, chunk: int = 2048) -> float:
    """meshProcess.py:18-35: mean squared nearest-neighbour distance a->b plus b->a on `point_num` samples per side
    (the reference samples mesh surfaces with Open3D; here the samples are drawn from the surface vertex sets).
    point_num=None uses every vertex (no sampling floor)."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def pick(p):
        if point_num is None or p.shape[0] <= point_num:
            return p
        return p[torch.randperm(p.shape[0], generator=g)[:point_num].to(p.device)]
    a, b = pick(pa).float(), pick(pb).float()
    if a.shape[0] == 0 or b.shape[0] == 0:
        return float("nan")

    def one_way(x, y):
        mins = []
        step = max(1, min(chunk, (1 << 28) // max(1, y.shape[0])))       # bound the distance block to ~1 GiB
        for i in range(0, x.shape[0], step):
            d = torch.cdist(x[i:i + step], y, compute_mode="donot_use_mm_for_euclid_dist")   # exact differences
            mins.append(d.min(dim=1).values)
        return float((torch.cat(mins) ** 2).mean())
    return one_way(b, a) + one_way(a, b)


_TET_TRI = torch.tensor([[-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4],
                         [3, 1, 5, -1, -1, -1], [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1],
                         [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1], [3, 2, 0, 3, 5, 2], [1, 3, 5, -1, -1, -1],
                         [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]])
_TET_NTRI = torch.tensor([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0])
_TET_EDGES = torch.tensor([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
_CUBE_TETS = torch.tensor([[0, 1, 3, 7], [0, 3, 2, 7], [0, 2, 6, 7], [0, 6, 4, 7], [0, 4, 5, 7], [0, 5, 1, 7]])


def marching_tetrahedra(volume: torch.Tensor, max_cells: int = 4_000_000):
    """Level-0 triangle mesh over the cells that straddle the surface (6 tetrahedra per cell, 16-case table).
    Returns (vertices [V,3] in grid coordinates, faces [F,3])."""
    dev = volume.device
    v = volume.float()
    occ = v > 0
    R = v.shape
    c = torch.zeros((R[0] - 1, R[1] - 1, R[2] - 1), dtype=torch.int32, device=dev)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                c += occ[dx:R[0] - 1 + dx, dy:R[1] - 1 + dy, dz:R[2] - 1 + dz].int()
    cells = ((c > 0) & (c < 8)).nonzero()
    if cells.shape[0] == 0:
        return torch.zeros((0, 3), device=dev), torch.zeros((0, 3), dtype=torch.long, device=dev)
    if cells.shape[0] > max_cells:
        raise RuntimeError(f"{cells.shape[0]} surface cells: volume is not a surface (noise?)")
    corner = torch.tensor([[i & 1, (i >> 1) & 1, (i >> 2) & 1] for i in range(8)], device=dev)
    cp = cells[:, None, :] + corner[None]                                   # [C,8,3] grid points
    lin = (cp[..., 0] * R[1] + cp[..., 1]) * R[2] + cp[..., 2]              # [C,8] linear ids
    tets = lin[:, _CUBE_TETS.to(dev)].reshape(-1, 4)                        # [6C,4]
    flat = v.reshape(-1)
    o = (flat[tets] > 0)
    code = (o.long() * torch.tensor([1, 2, 4, 8], device=dev)).sum(-1)
    keep = (code > 0) & (code < 15)
    tets, code = tets[keep], code[keep]
    e = tets[:, _TET_EDGES.to(dev)]                                         # [T,6,2]
    e = torch.sort(e, dim=-1).values
    tri = _TET_TRI.to(dev)[code]                                            # [T,6]
    ntri = _TET_NTRI.to(dev)[code]
    faces_e = []
    for k in range(2):
        m = ntri > k
        sel = tri[m][:, 3 * k:3 * k + 3]
        faces_e.append(torch.gather(e[m], 1, sel[..., None].expand(-1, -1, 2)))   # [F,3,2]
    fe = torch.cat(faces_e, dim=0).reshape(-1, 2)
    key = fe[:, 0] * flat.shape[0] + fe[:, 1]
    uniq, inv = torch.unique(key, return_inverse=True)
    a, b = uniq // flat.shape[0], uniq % flat.shape[0]
    va, vb = flat[a], flat[b]
    t = (va / (va - vb)).unsqueeze(-1)

    def coords(l):
        return torch.stack([l // (R[1] * R[2]), (l // R[2]) % R[1], l % R[2]], dim=-1).float()
    verts = coords(a) * (1 - t) + coords(b) * t
    return verts, inv.reshape(-1, 3)




_MC = None


def _mc_tables():
    """The 256-case table, rebuilt from its derivation (tools/make_mc_table.py: loop tracing on the cube with the
    'separate the inside corners' face rule).  PyMCubes' own table is not available here (parity unpinned)."""
    global _MC
    if _MC is None:
        import importlib.util
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_mc_table.py")
        spec = importlib.util.spec_from_file_location("make_mc_table", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        table = mod.build()
        tri = torch.full((256, 15), -1, dtype=torch.long)
        for b, t in enumerate(table):
            flat = [e for x in t for e in x]
            tri[b, :len(flat)] = torch.tensor(flat, dtype=torch.long)
        ntri = torch.tensor([len(t) for t in table])
        lo = torch.tensor([e[0] for e in mod.EDGES])
        axis = torch.tensor([e[2] for e in mod.EDGES])
        _MC = (tri, ntri, lo, axis)
    return _MC


def marching_cubes(volume: torch.Tensor, level: float = 0.0, max_cells: int = 4_000_000):
    """visualize.py:100 (`mcubes.marching_cubes(volume, 0)`) restated: one vertex per sign-changing grid edge at the linear
    crossing, shared by the cells around it; triangles from the 256-case table.  Returns (vertices [V,3] in grid
    coordinates, faces [F,3]); vertices are ordered by (owner voxel, axis x < y < z) like the device kernel's."""
    tri, ntri, elo, eaxis = _mc_tables()
    v = volume.float() - level
    R = v.shape
    occ = v > 0
    code = torch.zeros((R[0] - 1, R[1] - 1, R[2] - 1), dtype=torch.long)
    for c in range(8):
        dx, dy, dz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        code += occ[dx:R[0] - 1 + dx, dy:R[1] - 1 + dy, dz:R[2] - 1 + dz].long() << c
    cells = ((code > 0) & (code < 255)).nonzero()
    if cells.shape[0] == 0:
        return torch.zeros((0, 3)), torch.zeros((0, 3), dtype=torch.long)
    if cells.shape[0] > max_cells:
        raise RuntimeError(f"{cells.shape[0]} surface cells: volume is not a surface (noise?)")
    cc = code[cells[:, 0], cells[:, 1], cells[:, 2]]
    flat = v.reshape(-1)
    fe_lo, fe_ax = [], []
    for k in range(5):
        m = ntri[cc] > k
        e = tri[cc[m]][:, 3 * k:3 * k + 3]                                   # [F,3] cube edge ids
        lo = elo[e]                                                          # lower corner of each edge
        base = cells[m][:, None, :] + torch.stack([lo & 1, (lo >> 1) & 1, (lo >> 2) & 1], dim=-1)
        fe_lo.append((base[..., 0] * R[1] + base[..., 1]) * R[2] + base[..., 2])
        fe_ax.append(eaxis[e])
    lo_lin = torch.cat(fe_lo).reshape(-1)
    ax = torch.cat(fe_ax).reshape(-1)
    key = lo_lin * 3 + ax                                                    # (owner voxel, axis): the device's vertex order
    uniq, inv = torch.unique(key, return_inverse=True)
    a = uniq // 3
    axis = uniq % 3
    stride = torch.tensor([R[1] * R[2], R[2], 1])[axis]
    b = a + stride
    va, vb = flat[a], flat[b]
    t = va / (va - vb)
    pos = torch.stack([a // (R[1] * R[2]), (a // R[2]) % R[1], a % R[2]], dim=-1).float()
    pos[torch.arange(pos.shape[0]), axis] += t
    return pos, inv.reshape(-1, 3)


def smooth_simple(verts: torch.Tensor, faces: torch.Tensor, iterations: int = 10) -> torch.Tensor:
    """Open3D filter_smooth_simple (drag_utils.py:300): v <- (v + sum over adjacent vertices) / (1 + valence), Jacobi
    sweeps, adjacency = the unique vertex pairs of the faces."""
    v = verts.double().clone()
    e = torch.cat([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    e = torch.cat([e, e.flip(1)])
    e = torch.unique(e, dim=0)                                   # directed, each neighbour once
    n = torch.zeros(v.shape[0], dtype=torch.float64).index_add_(0, e[:, 0], torch.ones(e.shape[0], dtype=torch.float64))
    for _ in range(iterations):
        acc = torch.zeros_like(v).index_add_(0, e[:, 0], v[e[:, 1]])
        v = (v + acc) / (1.0 + n).unsqueeze(1)
    return v.float()


def mesh_occupancy(verts: torch.Tensor, faces: torch.Tensor, points: torch.Tensor, chunk: int = 4096) -> torch.Tensor:
    """Ray-parity occupancy (the device kernel's rule: crossings of p + s*(1,0,0), s > 0, counted when the point's (y,z)
    is strictly inside the triangle's (y,z) projection; barycentric form about a triangle vertex, edge-on triangles skipped).  Stands in for Open3D's RaycastingScene.compute_occupancy
    (drag_utils.py:437-440), which is absent here."""
    v = verts.float()
    A, B, C = v[faces[:, 0].long()], v[faces[:, 1].long()], v[faces[:, 2].long()]
    out = torch.empty(points.shape[0])
    for i in range(0, points.shape[0], chunk):
        p = points[i:i + chunk].float()
        uy, uz = (B[:, 1] - A[:, 1])[None], (B[:, 2] - A[:, 2])[None]
        wy, wz = (C[:, 1] - A[:, 1])[None], (C[:, 2] - A[:, 2])[None]
        qy, qz = p[:, None, 1] - A[None, :, 1], p[:, None, 2] - A[None, :, 2]
        D = uy * wz - uz * wy
        ok = D.abs() > 1e-6 * (uy.abs() + uz.abs()) * (wy.abs() + wz.abs())       # edge-on triangles cannot be crossed
        Ds = torch.where(ok, D, torch.ones_like(D))
        sB, sC = (qy * wz - qz * wy) / Ds, (uy * qz - uz * qy) / Ds
        hit = ok & (sB > 0) & (sC > 0) & (sB + sC < 1)
        xh = A[None, :, 0] + sB * (B[:, 0] - A[:, 0])[None] + sC * (C[:, 0] - A[:, 0])[None]
        cross = hit & (xh > p[:, None, 0])
        out[i:i + chunk] = (cross.sum(dim=1) % 2).float()
    return out
