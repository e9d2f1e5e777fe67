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
