"""Mesh side of get_mesh (reference: triplane_decoder/visualize.py:100-104 = PyMCubes marching cubes at level 0 +
vertices/res*2-1; drag_utils.py:300 = Open3D filter_smooth_simple(10); meshProcess.py:18-35 = Chamfer distance).

Those are third-party CPU calls outside the kernel path.  Here the surface is produced ON THE
DEVICE by csrc/surface.hip (marching cubes on the resident volume, smoothing, nearest-neighbour Chamfer on area-uniform
surface samples) through the C ABI (ishap_surface_count / _emit, ishap_mesh_smooth, ishap_chamfer); this module only
allocates the outputs.  The backend is an explicit choice, never an import probe: BACKEND = "device" (default) or
"third_party" (the reference's own PyMCubes / Open3D calls on the host, for users who have them and want that exact mesh).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


MAX_OBJ_VERTICES = 4_000_000      # text export limit (a 256^3 shape has well under a million surface vertices)
BACKEND = "device"                # "device" | "open3d" | "third_party"; set explicitly, results never depend on what happens to be installed
METHODS = {"marching_tetrahedra": 0, "marching_cubes": 1}


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} runs on the GPU (libishap_hip.so); there is no CPU fallback")


def extract_surface(volume: torch.Tensor, level: float = 0.0, method: str = "marching_cubes"):
    """Level-`level` triangle mesh of a [res,res,res] volume: (vertices [V,3] float32 in grid coordinates,
    triangles [F,3] int32), both on the device, in voxel order (deterministic).  method: "marching_cubes" (what the
    reference calls, visualize.py:100) or "marching_tetrahedra"."""
    meth = METHODS[method]
    _need_gpu(volume, "extract_surface")
    assert volume.dim() == 3 and volume.shape[0] == volume.shape[1] == volume.shape[2]
    res = volume.shape[0]
    dev = volume.device
    vol = volume.detach().to(torch.float32).contiguous()
    L = _lib.lib()
    scratch = torch.empty(int(L.ishap_surface_scratch_bytes(res)), dtype=torch.uint8, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        s = _lib.stream_ptr(dev)
        _lib.check(L.ishap_surface_count(vol.data_ptr(), res, float(level), meth, scratch.data_ptr(), counts.data_ptr(), s))
        nv, nt = (int(c) for c in counts.tolist())                 # the one host read-back: output sizes
        verts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        tris = torch.empty((nt, 3), dtype=torch.int32, device=dev)
        if nv and nt:
            _lib.check(L.ishap_surface_emit(vol.data_ptr(), res, float(level), meth, scratch.data_ptr(), verts.data_ptr(),
                                            tris.data_ptr(), s))
    return verts, tris


def surface_counts(volume: torch.Tensor, level: float = 0.0, method: str = "marching_cubes"):
    """(vertices, triangles) of the level surface without emitting it."""
    meth = METHODS[method]
    _need_gpu(volume, "surface_counts")
    res = volume.shape[0]
    dev = volume.device
    vol = volume.detach().to(torch.float32).contiguous()
    L = _lib.lib()
    scratch = torch.empty(int(L.ishap_surface_scratch_bytes(res)), dtype=torch.uint8, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.ishap_surface_count(vol.data_ptr(), res, float(level), meth, scratch.data_ptr(), counts.data_ptr(),
                                         _lib.stream_ptr(dev)))
    nv, nt = counts.tolist()
    return int(nv), int(nt)


def smooth_mesh(verts: torch.Tensor, tris: torch.Tensor, iterations: int = 10, box_max: float = 0.0) -> torch.Tensor:
    """filter_smooth_simple (drag_utils.py:300) on the device; returns new vertex positions.  box_max > 0: `verts` are grid
    coordinates of a [0, box_max]^3 volume whose surface may be cut open by the box (each neighbour still counts once)."""
    _need_gpu(verts, "smooth_mesh")
    out = verts.detach().to(torch.float32).contiguous().clone()
    tris = tris.detach().to(torch.int32).contiguous()
    if out.shape[0] == 0 or tris.shape[0] == 0 or iterations <= 0:
        return out
    nbytes = int(_lib.lib().ishap_mesh_smooth_scratch_bytes(out.shape[0], tris.shape[0]))
    if nbytes < 0:
        raise RuntimeError("ishap_mesh_smooth_scratch_bytes: invalid sizes")
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().ishap_mesh_smooth(out.data_ptr(), out.shape[0], tris.data_ptr(), tris.shape[0], int(iterations),
                                                float(box_max), scratch.data_ptr(), nbytes, _lib.stream_ptr(out.device)))
    return out


def chamfer_distance(pa: torch.Tensor, pb: torch.Tensor, point_num=20000, seed: int = 0, query_num=None) -> float:
    """meshProcess.py:18-35: mean squared nearest-neighbour distance a->b plus b->a on `point_num` samples per side
    (the reference samples mesh surfaces with Open3D; here the samples are drawn from the given point sets).
    point_num=None uses every point (no sampling floor).  query_num (with point_num=None): only the QUERY side of each
    direction is a random subset of that many points, the nearest neighbour is searched in the COMPLETE other set -- an
    unbiased estimate of the all-points value without its quadratic cost and without a sampling floor (the floor of the
    sampled form comes from thinning the target set)."""
    _need_gpu(pa, "chamfer_distance")
    g = torch.Generator(device="cpu").manual_seed(seed)

    def pick(p, n):
        if n is None or p.shape[0] <= n:
            return p
        return p[torch.randperm(p.shape[0], generator=g)[:n].to(p.device)]

    def run(a, b):
        nearest = torch.empty(max(a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
        out2 = torch.empty(2, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().ishap_chamfer(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], nearest.data_ptr(),
                                                out2.data_ptr(), _lib.stream_ptr(a.device)))
        return out2.tolist()
    a = pick(pa, point_num).detach().to(torch.float32).contiguous()
    b = pick(pb, point_num).detach().to(device=a.device, dtype=torch.float32).contiguous()
    if a.shape[0] == 0 or b.shape[0] == 0:
        return float("nan")
    if query_num is not None and point_num is None:
        return float(run(pick(a, query_num).contiguous(), b)[0] + run(pick(b, query_num).contiguous(), a)[0])
    d = run(a, b)
    return float(d[0] + d[1])


def mesh_chamfer(mesh_a, mesh_b, point_num: int = 20000, seed: int = 0) -> float:
    """calc_chamfer (meshProcess.py:18-35): `point_num` points sampled uniformly BY AREA on each surface
    (sample_points_uniformly), then the two mean squared nearest-neighbour distances.  mesh_*: (vertices, triangles) pairs
    or OccupancyMesh objects, on the device."""
    def vt(m):
        return (m.vertices, m.triangles) if isinstance(m, OccupancyMesh) else m
    (va, ta), (vb, tb) = vt(mesh_a), vt(mesh_b)
    g = torch.Generator(device="cpu").manual_seed(seed)
    pa = sample_surface_points(va, ta, point_num, g)
    pb = sample_surface_points(vb, tb, point_num, g)
    return chamfer_distance(pa, pb, point_num=None)


class _Points:
    """What `np.asarray(mesh.vertices)` / `len(mesh.vertices)` need (main.py:507): a lazy host view of a device array."""

    def __init__(self, fetch):
        self._fetch = fetch

    def __array__(self, dtype=None, copy=None):
        a = self._fetch()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return int(self._fetch().shape[0])


class OccupancyMesh:
    """The mesh get_mesh returns with the default "device" backend: the device volume plus, on first use, its surface
    (vertices in the reference's convention grid/res*2-1, visualize.py:101) after `smooth_iterations` sweeps.

    It carries the part of open3d.geometry.TriangleMesh's surface the reference's GUI touches on `drag_stuff.mesh`
    (main.py:288,314,373,478,507) -- `has_vertex_normals()`, `has_triangle_normals()`, `compute_vertex_normals()`,
    `np.asarray(mesh.vertex_normals)`, `copy.deepcopy` -- and `to_open3d()`, which builds the real TriangleMesh for
    `update_mesh` (Open3D is imported only there, when called).  `.vertices` / `.triangles` stay device tensors (the
    product's own consumers: Chamfer, sampling, OBJ export); `np.asarray` of them works through torch's __array__ only for
    host tensors, so GUI code goes through `to_open3d()` or `vertices_numpy()`."""

    def __init__(self, volume: torch.Tensor, res: int, smooth_iterations: int = 10):
        self.volume = volume
        self.res = res
        self.smooth_iterations = smooth_iterations
        self._mesh = None
        self._vnormals = None

    def _build(self):
        if self._mesh is None:
            v, t = extract_surface(self.volume, 0.0)
            v = smooth_mesh(v, t, self.smooth_iterations, box_max=float(self.res - 1))     # in grid coordinates: the box is known
            self._mesh = (v / self.res * 2 - 1, t)
        return self._mesh

    @property
    def vertices(self) -> torch.Tensor:
        return self._build()[0]

    @property
    def triangles(self) -> torch.Tensor:
        return self._build()[1]

    def counts(self):
        return surface_counts(self.volume, 0.0)

    # ---- the TriangleMesh surface main.py uses ----
    def vertices_numpy(self) -> np.ndarray:
        return self.vertices.detach().cpu().numpy().astype(np.float64)

    def triangles_numpy(self) -> np.ndarray:
        return self.triangles.detach().cpu().numpy().astype(np.int32)

    def has_vertex_normals(self) -> bool:
        return self._vnormals is not None

    def has_triangle_normals(self) -> bool:
        return False

    def compute_vertex_normals(self, normalized: bool = True):
        """TriangleMesh.compute_vertex_normals: area-weighted sum of the incident triangles' normals, normalised."""
        self._vnormals = vertex_normals(self.vertices, self.triangles, normalized)
        return self

    @property
    def vertex_normals(self):
        return _Points(lambda: np.zeros((0, 3)) if self._vnormals is None else self._vnormals.detach().cpu().numpy().astype(np.float64))

    def to_open3d(self):
        """The open3d.geometry.TriangleMesh the reference's get_mesh returns (visualize.py:102-104 + drag_utils.py:300):
        same vertices / triangles, already smoothed on the device.  Imports Open3D here and nowhere else."""
        import open3d as o3d
        m = o3d.geometry.TriangleMesh()
        m.vertices = o3d.utility.Vector3dVector(self.vertices_numpy())
        m.triangles = o3d.utility.Vector3iVector(self.triangles_numpy())
        if self._vnormals is not None:
            m.vertex_normals = o3d.utility.Vector3dVector(np.asarray(self.vertex_normals))
        return m

    def __deepcopy__(self, memo):
        m = OccupancyMesh(self.volume.clone(), self.res, self.smooth_iterations)
        if self._mesh is not None:
            m._mesh = (self._mesh[0].clone(), self._mesh[1].clone())
        if self._vnormals is not None:
            m._vnormals = self._vnormals.clone()
        return m


def vertex_normals(verts: torch.Tensor, tris: torch.Tensor, normalized: bool = True) -> torch.Tensor:
    """Per-vertex normals as Open3D's compute_vertex_normals defines them: the sum over incident triangles of the
    (unnormalised, i.e. area-weighted) face normal cross(v1 - v0, v2 - v0), then normalised.  Display-side plumbing in
    torch ops on the device (not on the timed path)."""
    v = verts.detach().to(torch.float32)
    t = tris.detach().to(device=v.device, dtype=torch.long)
    fn = torch.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]], dim=1)
    out = torch.zeros_like(v)
    for k in range(3):
        out.index_add_(0, t[:, k], fn)
    if normalized:
        out = out / out.norm(dim=1, keepdim=True).clamp_min(1e-30)
    return out


def mesh_arrays(mesh):
    """(vertices [V,3] float32, triangles [F,3] int32) as numpy from whatever a caller hands train_triplane(mesh=...):
    a (vertices, triangles) pair, an OccupancyMesh, or any object with `.vertices` / `.triangles` (an
    open3d.geometry.TriangleMesh, main.py:447-451) -- read with np.asarray, nothing Open3D-specific."""
    if isinstance(mesh, OccupancyMesh):
        return mesh.vertices_numpy().astype(np.float32), mesh.triangles_numpy()
    if isinstance(mesh, tuple):
        v, t = mesh
    else:
        v, t = mesh.vertices, mesh.triangles
    if torch.is_tensor(v):
        v = v.detach().cpu().numpy()
    if torch.is_tensor(t):
        t = t.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float32).reshape(-1, 3), np.asarray(t, dtype=np.int32).reshape(-1, 3)


def volume_to_mesh(volume: torch.Tensor, res: int, smooth_iterations: int = 10, backend: str = None):
    """get_mesh's mesh (drag_utils.py:298-300).  backend None -> the module-level BACKEND:
      "device"      OccupancyMesh (surface, smoothing on the device; vertices / triangles stay device tensors)
      "open3d"      the same device surface handed over as an open3d.geometry.TriangleMesh -- what the reference's GUI
                    expects from drag_stuff.mesh (main.py:288,314,373,478,507); only the final vertex / triangle arrays
                    cross PCIe (a few MB, not the 67 MB volume)
      "third_party" the reference's own PyMCubes + Open3D calls on the host."""
    backend = BACKEND if backend is None else backend
    if backend == "device":
        return OccupancyMesh(volume, res, smooth_iterations)
    if backend == "open3d":
        return OccupancyMesh(volume, res, smooth_iterations).to_open3d()
    if backend != "third_party":
        raise ValueError(f"unknown mesh backend {backend!r}")
    import mcubes
    import open3d as o3d
    vertices, triangles = mcubes.marching_cubes(volume.detach().cpu().numpy(), 0)
    vertices = vertices / res * 2 - 1                      # visualize.py:101 (create_obj_o3d's own convention)
    mesh = o3d.geometry.TriangleMesh()
    mesh.vertices = o3d.utility.Vector3dVector(vertices)
    mesh.triangles = o3d.utility.Vector3iVector(triangles)
    return mesh.filter_smooth_simple(number_of_iterations=smooth_iterations) if smooth_iterations > 0 else mesh


def _write_obj(path, verts: torch.Tensor, tris: torch.Tensor):
    v = verts.detach().cpu().numpy()
    f = tris.detach().cpu().numpy() + 1
    with open(path, "w") as fh:
        fh.write("".join(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n" for p in v))
        fh.write("".join(f"f {t[0]} {t[1]} {t[2]}\n" for t in f))


def write_mesh(path, mesh):
    """o3d.io.write_triangle_mesh (drag_utils.py:470) when the mesh is an Open3D object; Wavefront OBJ of the device mesh otherwise."""
    if isinstance(mesh, OccupancyMesh):
        nv, nt = mesh.counts()
        if nv > MAX_OBJ_VERTICES:
            with open(path, "w") as f:
                f.write(f"# {nv} vertices / {nt} triangles: the volume is not a surface, mesh not written\n")
            return
        _write_obj(path, mesh.vertices, mesh.triangles)
        return
    import open3d as o3d
    o3d.io.write_triangle_mesh(path, mesh)


def read_obj(path: str):
    """Minimal Wavefront OBJ reader (v / f records, polygons fan-triangulated): (vertices [V,3] float32, triangles [F,3] int32)."""
    vs, fs = [], []
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                vs.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "f":
                idx = [int(t.split("/")[0]) for t in p[1:]]
                idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
                fs.extend([idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1))
    return np.asarray(vs, np.float32).reshape(-1, 3), np.asarray(fs, np.int32).reshape(-1, 3)


def mesh_occupancy(verts: torch.Tensor, tris: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """RaycastingScene.compute_occupancy of the reference (drag_utils.py:437-440) on the device: 1 inside / 0 outside the
    closed triangle mesh, by ray parity.  verts [V,3], tris [F,3], points [P,3] -> [P] float32."""
    _need_gpu(verts, "mesh_occupancy")
    dev = verts.device
    v = verts.detach().to(torch.float32).contiguous()
    t = tris.detach().to(device=dev, dtype=torch.int32).contiguous()
    p = points.detach().to(device=dev, dtype=torch.float32).contiguous()
    occ = torch.empty(p.shape[0], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ishap_mesh_occupancy(v.data_ptr(), t.data_ptr(), t.shape[0], p.data_ptr(), p.shape[0], occ.data_ptr(),
                                                   _lib.stream_ptr(dev)))
    return occ


def sample_surface_points(verts: torch.Tensor, tris: torch.Tensor, n: int, generator=None,
                          multinomial_max: int = 1 << 24) -> torch.Tensor:
    """mesh.sample_points_uniformly(n) (drag_utils.py:432): triangles drawn with probability proportional to their area,
    a uniform point on each.  The random draws are torch's (plumbing); areas and points are computed by the library.
    Above `multinomial_max` triangles (torch.multinomial's category limit, 2^24) the triangle is drawn by inverse CDF instead:
    the same distribution from ONE uniform per sample, so the samples of a given seed differ between the two branches
    (only meshes beyond 16.7 M triangles -- random-weight noise surfaces -- take the second)."""
    _need_gpu(verts, "sample_surface_points")
    dev = verts.device
    v = verts.detach().to(torch.float32).contiguous()
    t = tris.detach().to(device=dev, dtype=torch.int32).contiguous()
    areas = torch.empty(t.shape[0], dtype=torch.float32, device=dev)
    L = _lib.lib()
    with torch.cuda.device(dev):
        _lib.check(L.ishap_mesh_tri_areas(v.data_ptr(), t.data_ptr(), t.shape[0], areas.data_ptr(), _lib.stream_ptr(dev)))
        if areas.numel() <= multinomial_max:
            idx = torch.multinomial(areas.double().cpu(), n, replacement=True, generator=generator).to(device=dev, dtype=torch.int32)
        else:      # torch.multinomial stops at 2^24 categories (a 256^3 noise surface has 3e7 triangles): inverse CDF instead
            cdf = torch.cumsum(areas.double().cpu(), 0)
            u = torch.rand(n, generator=generator, dtype=torch.float64) * cdf[-1]
            # right=True: triangle i owns [cdf[i-1], cdf[i]) -- a draw that lands ON a cdf value (u = 0 in front of leading
            # zero-area triangles included) goes to the next triangle with area, never to a degenerate one
            idx = torch.searchsorted(cdf, u, right=True).clamp_(max=areas.numel() - 1).to(device=dev, dtype=torch.int32)
        uw = torch.rand((n, 2), generator=generator).to(dev).contiguous()
        pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
        _lib.check(L.ishap_mesh_points_on_tris(v.data_ptr(), t.data_ptr(), idx.data_ptr(), uw.data_ptr(), n, pts.data_ptr(),
                                               _lib.stream_ptr(dev)))
    return pts


def sample_occupancy(mesh, mesh_path, center_mesh, points_size, uniform_ratio, device=None, generator=None):
    """drag_utils.py:411-440: `points_size` samples (a `uniform_ratio` share uniform in [-1,1]^3, the rest on the surface
    plus N(0, 0.01) noise) with their occupancy.  BACKEND "third_party": Open3D (RaycastingScene) exactly as the
    reference; "device" (default): the mesh -- an OBJ file, a (vertices, triangles) pair, an OccupancyMesh or ANY object
    with `.vertices` / `.triangles` such as the open3d TriangleMesh the GUI passes (main.py:447-451) -- is sampled on
    the device.  Returns (None, None) when no mesh is given."""
    if mesh is None and mesh_path is None:
        return None, None
    o3d = None
    if BACKEND == "third_party":
        import open3d as o3d
    if o3d is not None and not isinstance(mesh, tuple):
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
    # ---- device route ----
    if mesh is not None:
        v_np, t_np = mesh_arrays(mesh)
    else:
        v_np, t_np = read_obj(mesh_path)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    v = torch.from_numpy(v_np).to(dev)
    t = torch.from_numpy(t_np).to(dev)
    if center_mesh:                                              # drag_utils.py:420-428
        mx, mn = v.max(dim=0).values, v.min(dim=0).values
        if bool((mn > 1).any() or (mn < -1).any() or (mx > 1).any() or (mx < -1).any()):
            v = v - v.mean(dim=0)                                # get_center() = mean of the vertices
            ext = float((mx - mn).max())
            if ext > 2:
                v = v * (2. / (ext + 1e-2))
    n_uniform = int(points_size * uniform_ratio)
    uniform = (torch.rand((n_uniform, 3), generator=generator) * 2 - 1).to(dev)
    surf = sample_surface_points(v, t, points_size - n_uniform, generator)
    surf = surf + 0.01 * torch.randn(surf.shape, generator=generator).to(dev)
    pts = torch.cat([uniform, surf], dim=0).contiguous()
    occ = mesh_occupancy(v, t, pts)
    return pts.cpu().numpy(), occ.reshape(-1, 1).cpu().numpy()


def export_obj(volume: torch.Tensor, path: str, scale_div: float = 255.0, backend: str = None):
    """visualize.py:71-73 (create_obj): surface at 0, vertices / 255 * 2 - 1, Wavefront OBJ."""
    backend = BACKEND if backend is None else backend
    if backend == "third_party":
        import mcubes
        vertices, triangles = mcubes.marching_cubes(volume.detach().cpu().numpy(), 0)
        vertices = vertices / scale_div * 2 - 1
        mcubes.export_obj(vertices, triangles, path)
        return
    nv, nt = surface_counts(volume, 0.0)
    if nv > MAX_OBJ_VERTICES:                      # not a surface (e.g. random weights decode to noise): do not write GBs of text
        with open(path, "w") as f:
            f.write(f"# {nv} vertices / {nt} triangles: the volume is not a surface, mesh not written\n")
        return
    verts, tris = extract_surface(volume, 0.0)
    _write_obj(path, verts / scale_div * 2 - 1, tris)
