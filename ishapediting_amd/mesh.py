"""Mesh extraction side of get_mesh (reference: triplane_decoder/visualize.py:100-104 = PyMCubes
marching cubes at level 0 + vertices/res*2-1; drag_utils.py:300 = Open3D filter_smooth_simple(10)).

Both are third-party CPU code outside the kernel path ("next").  When PyMCubes
and Open3D are importable the reference's exact calls are used; otherwise the decoded volume is
returned wrapped in `OccupancyMesh`, which keeps the device volume and exposes the surface-crossing
voxel count so callers (bench, tests) still have a size-independent summary.
"""
from __future__ import annotations

import copy

import numpy as np
import torch


class OccupancyMesh:
    """Stand-in mesh when PyMCubes/Open3D are absent: the occupancy-logit volume itself."""

    def __init__(self, volume: torch.Tensor, res: int):
        self.volume = volume
        self.res = res

    def occupied_voxels(self) -> int:
        return int((self.volume > 0).sum().item())

    def surface_cells(self) -> int:
        """Number of grid cells whose 8 corners straddle level 0 (what marching cubes would triangulate)."""
        o = self.volume > 0
        c = o[:-1, :-1, :-1].int()
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    if dx or dy or dz:
                        c = c + o[dx:o.shape[0] - 1 + dx, dy:o.shape[1] - 1 + dy, dz:o.shape[2] - 1 + dz].int()
        return int(((c > 0) & (c < 8)).sum().item())

    def __deepcopy__(self, memo):
        return OccupancyMesh(self.volume.clone(), self.res)


def volume_to_mesh(volume: torch.Tensor, res: int, smooth_iterations: int = 10):
    try:
        import mcubes           # noqa: F401
        import open3d as o3d    # noqa: F401
    except Exception:
        return OccupancyMesh(volume, res)
    vertices, triangles = mcubes.marching_cubes(volume.detach().cpu().numpy(), 0)
    vertices = vertices / res * 2 - 1                      # visualize.py:101 (create_obj_o3d's own convention)
    mesh = o3d.geometry.TriangleMesh()
    mesh.vertices = o3d.utility.Vector3dVector(vertices)
    mesh.triangles = o3d.utility.Vector3iVector(triangles)
    return mesh.filter_smooth_simple(number_of_iterations=smooth_iterations)


def write_mesh(path, mesh):
    """o3d.io.write_triangle_mesh (drag_utils.py:470) when Open3D produced the mesh; the raw volume otherwise."""
    if isinstance(mesh, OccupancyMesh):
        np.save(path + ".volume.npy", mesh.volume.detach().cpu().numpy())
        return
    import open3d as o3d
    o3d.io.write_triangle_mesh(path, mesh)


def sample_occupancy(mesh, mesh_path, center_mesh, points_size, uniform_ratio):
    """drag_utils.py:411-440 via Open3D (RaycastingScene); returns (None, None) when no mesh is given."""
    if mesh is None and mesh_path is None:
        return None, None
    import open3d as o3d        # required for this route, exactly as in the reference
    if mesh is None:
        mesh = o3d.io.read_triangle_mesh(mesh_path)
    if center_mesh:
        max_bound, min_bound = mesh.get_max_bound(), mesh.get_min_bound()
        axis_extent = max_bound - min_bound
        if np.any(min_bound > 1) or np.any(min_bound < -1) or np.any(max_bound > 1) or np.any(max_bound < -1):
            mesh.translate(-mesh.get_center())
            if axis_extent.max() > 2:
                mesh.scale(2. / (axis_extent.max() + 1e-2), center=np.array([0., 0, 0]))
    n_uniform = int(points_size * uniform_ratio)
    uniform = (np.random.rand(n_uniform, 3) * 2 - 1).astype(np.float32)
    surf = np.asarray(mesh.sample_points_uniformly(points_size - n_uniform).points, dtype=np.float32)
    surf += 0.01 * np.random.randn(surf.shape[0], 3)
    pts = np.concatenate([uniform, surf], axis=0).astype(np.float32)
    scene = o3d.t.geometry.RaycastingScene()
    scene.add_triangles(o3d.t.geometry.TriangleMesh().from_legacy(mesh_legacy=mesh))
    occ = scene.compute_occupancy(pts).numpy().reshape(-1, 1).astype(np.float32)
    return pts, occ


# ------------------------------------------------------------------------------------------------------------
# level-0 surface without PyMCubes
# ------------------------------------------------------------------------------------------------------------
def mc_vertices(volume: torch.Tensor, level: float = 0.0) -> torch.Tensor:
    """The vertex set marching cubes produces: one vertex per grid edge whose end points straddle `level`,
    at the linearly interpolated crossing (grid coordinates, [V,3]).  Triangulation tables are not needed for
    vertex counts or for the Chamfer metric of meshProcess.py:18-35."""
    v = volume.float() - level
    out = []
    for axis in range(3):
        a = v.narrow(axis, 0, v.shape[axis] - 1)
        b = v.narrow(axis, 1, v.shape[axis] - 1)
        cross = (a < 0) != (b < 0)
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


def export_obj(volume: torch.Tensor, path: str, scale_div: float = 255.0):
    """visualize.py:71-73 (create_obj): marching cubes at 0, vertices / 255 * 2 - 1, Wavefront OBJ.
    PyMCubes when importable (the reference's exact call); marching tetrahedra otherwise."""
    try:
        import mcubes
        vertices, triangles = mcubes.marching_cubes(volume.detach().cpu().numpy(), 0)
        vertices = vertices / scale_div * 2 - 1
        mcubes.export_obj(vertices, triangles, path)
        return
    except ImportError:
        pass
    try:
        verts, faces = marching_tetrahedra(volume)
    except RuntimeError as e:
        with open(path, "w") as f:
            f.write(f"# {e}\n")
        return
    verts = (verts / scale_div * 2 - 1).cpu().numpy()
    faces = faces.cpu().numpy() + 1
    with open(path, "w") as f:
        for v in verts:
            f.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for t in faces:
            f.write(f"f {t[0]} {t[1]} {t[2]}\n")
