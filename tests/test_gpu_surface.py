"""csrc/surface.hip (marching cubes / marching tetrahedra, smoothing, Chamfer on the device) against the CPU statement of
the same algorithms in oracle/surface_cpu.py, through the C ABI.  Parity with the reference's third-party calls (PyMCubes /
Open3D) is unpinned (absent here, versions unpinned upstream); what is pinned is the algorithm stated in the oracle
plus analytic properties at the full 256^3 size."""
import numpy as np
import pytest
import torch

from oracle import surface_cpu as S

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def sphere(res, r, centre=None):
    ax = torch.arange(res, dtype=torch.float32) - ((res - 1) / 2 if centre is None else centre)
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    return r - torch.sqrt(x * x + y * y + z * z)


def smooth_field(res, seed):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn((1, 1, 6, 6, 6), generator=g)
    return torch.nn.functional.interpolate(f, size=(res, res, res), mode="trilinear", align_corners=True)[0, 0].contiguous()


def canon_faces(faces, index_map=None):
    """index triples rotated so the smallest index comes first (orientation kept), optionally re-indexed."""
    out = []
    for t in faces.cpu().numpy().tolist():
        if index_map is not None:
            t = [index_map[i] for i in t]
        k = t.index(min(t))
        out.append((t[k], t[(k + 1) % 3], t[(k + 2) % 3]))
    return sorted(out)


def torus(res, R, r):
    ax = torch.arange(res, dtype=torch.float32) - (res - 1) / 2
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    return r - torch.sqrt((torch.sqrt(x * x + y * y) - R) ** 2 + z * z)


@pytest.mark.parametrize("method", ["marching_cubes", "marching_tetrahedra"])
@pytest.mark.parametrize("case", ["sphere", "field", "noise", "ties", "empty", "full"])
def test_surface_matches_cpu_statement(case, method):
    from ishapediting_amd.mesh import extract_surface, surface_counts
    res = 24
    vol = {"sphere": sphere(res, 7.3), "field": smooth_field(res, 3), "empty": -torch.ones((res,) * 3),
           "full": torch.ones((res,) * 3),
           "noise": torch.randn((res,) * 3, generator=torch.Generator().manual_seed(8)),   # every ambiguous case
           # integer-valued: a third of the voxels sit exactly AT the level (tie rule: `value <= level` vs `value > level`,
           # the cut PyMCubes' published source makes; oracle/surface_cpu.py:mc_vertices)
           "ties": torch.randint(-1, 2, (res,) * 3, generator=torch.Generator().manual_seed(9)).float()}[case]
    v_d, f_d = extract_surface(vol.to(dev()), method=method)
    if case in ("empty", "full"):
        assert v_d.shape == (0, 3) and f_d.shape == (0, 3) and surface_counts(vol.to(dev()), method=method) == (0, 0)
        return
    v_o, f_o = S.marching_cubes(vol) if method == "marching_cubes" else S.marching_tetrahedra(vol)
    assert v_d.shape == v_o.shape and f_d.shape == f_o.shape
    assert surface_counts(vol.to(dev()), method=method) == (v_o.shape[0], f_o.shape[0])
    if method == "marching_cubes":
        # the quantity north_star names: the marching-cubes vertex count = the sign-changing grid edges
        assert v_d.shape[0] == S.mc_vertices(vol).shape[0]
    if case == "ties":
        # vertices of the edges that leave an at-level voxel coincide (t = 0), so nearest-neighbour matching is not a
        # bijection here.  Values are -1 / 0 / 1, so every coordinate is an exact multiple of 0.5: the vertex MULTISETS must
        # be equal; marching cubes also emits in the statement's order (owner voxel, then axis): same faces index by index
        def rows(v):
            return sorted(map(tuple, v.cpu().tolist()))
        assert rows(v_d) == rows(v_o)
        if method == "marching_cubes":
            assert float((v_d.cpu() - v_o).abs().max()) == 0.0
            assert canon_faces(f_d) == canon_faces(f_o)
        at_corner = (v_o == v_o.round()).all(dim=1)
        assert int(at_corner.sum()) > 100          # the case does exercise the tie
        return
    # same vertex set: every device vertex has exactly one oracle vertex within 2e-5 (both are one fp32 interpolation
    # of the same two samples), and the map is a bijection
    d = torch.cdist(v_d.cpu().double(), v_o.double())
    near = d.argmin(dim=1)
    assert float(d.min(dim=1).values.max()) <= 2e-5
    assert torch.unique(near).numel() == v_o.shape[0]
    # same triangles with the same orientation, after mapping device indices to oracle indices
    assert canon_faces(f_d, near.tolist()) == canon_faces(f_o)
    # every vertex referenced, indices in range
    assert int(f_d.min()) == 0 and int(f_d.max()) == v_d.shape[0] - 1
    assert torch.unique(f_d).numel() == v_d.shape[0]


def test_smoothing_and_chamfer_match_cpu_statement():
    from ishapediting_amd.mesh import chamfer_distance, extract_surface, smooth_mesh
    vol = sphere(32, 9.3)
    v, f = extract_surface(vol.to(dev()))
    sm_d = smooth_mesh(v, f, 10)
    sm_o = S.smooth_simple(v.cpu(), f.cpu().long(), 10)
    assert float((sm_d.cpu() - sm_o).abs().max()) < 1e-4               # closed surface: per-face gathering = unique adjacency
    assert float((sm_d - v).abs().max()) > 1e-3                         # and it did move
    sm_d2 = smooth_mesh(v, f, 10)
    assert torch.equal(sm_d, sm_d2)                                    # fixed-point accumulation: bitwise repeatable
    # the sweeps ping-pong between the vertex buffer and scratch: odd counts end in scratch and are copied back; the adjacency
    # lists' internal order depends on atomics and must not matter; a mesh whose scan spans several 2048-entry blocks
    for iters in (1, 3):
        assert float((smooth_mesh(v, f, iters).cpu() - S.smooth_simple(v.cpu(), f.cpu().long(), iters)).abs().max()) < 1e-4, iters
    assert torch.equal(smooth_mesh(v, f, 0), v)
    vb, fb = extract_surface(sphere(64, 21.7).to(dev()))
    assert vb.shape[0] > 3 * 2048
    assert float((smooth_mesh(vb, fb, 3).cpu() - S.smooth_simple(vb.cpu(), fb.cpu().long(), 3)).abs().max()) < 1e-4
    pa, pb = v, v + torch.tensor([0.5, 0.0, 0.0], device=v.device)
    c_d = chamfer_distance(pa, pb, None)
    c_o = S.chamfer_distance(pa.cpu(), pb.cpu(), None)
    assert abs(c_d - c_o) <= 1e-5 * max(1.0, c_o)
    assert chamfer_distance(pa, pa.clone(), None) == 0.0
    c_s = chamfer_distance(pa, pb, 2000, seed=4)
    assert abs(c_s - S.chamfer_distance(pa.cpu(), pb.cpu(), 2000, seed=4)) <= 1e-5 * max(1.0, c_o)


def test_smoothing_rejects_an_undersized_scratch_buffer():
    """ABI version 2: the scratch size travels with the pointer; a buffer sized by the old 32-bytes-per-vertex rule
    fails the call instead of being written out of bounds."""
    from ishapediting_amd import _lib
    from ishapediting_amd.mesh import extract_surface
    v, f = extract_surface(sphere(32, 9.3).to(dev()))
    L = _lib.lib()
    assert L.ishap_version() >= 2
    need = int(L.ishap_mesh_smooth_scratch_bytes(v.shape[0], f.shape[0]))
    old_rule = 32 * v.shape[0]
    assert need > old_rule
    out = v.clone().contiguous()
    scratch = torch.empty(need, dtype=torch.uint8, device=v.device)
    with torch.cuda.device(v.device):
        rc = L.ishap_mesh_smooth(out.data_ptr(), v.shape[0], f.data_ptr(), f.shape[0], 2, 0.0, scratch.data_ptr(), old_rule,
                                 _lib.stream_ptr(v.device))
    assert rc != 0 and b"scratch" in L.ishap_last_error()
    torch.cuda.synchronize()
    assert torch.equal(out, v)                                          # nothing ran


def test_area_sampling_by_inverse_cdf_never_picks_a_degenerate_triangle():
    """The branch sample_surface_points takes above torch.multinomial's 2^24 categories, forced here by its threshold parameter:
    zero-area triangles (leading ones included, where a draw of exactly 0 would land) are never chosen, and the draw is
    area-proportional."""
    from ishapediting_amd.mesh import sample_surface_points
    # a unit square in z = 0 (two triangles), a far square of 3x the area, and degenerate triangles first, between and last
    v = torch.tensor([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [10, 0, 0], [13, 0, 0], [13, 1, 0], [10, 1, 0], [5, 5, 5]],
                     dtype=torch.float32, device=dev())
    f = torch.tensor([[8, 8, 8], [0, 0, 1], [0, 1, 2], [0, 2, 3], [8, 8, 8], [4, 5, 6], [4, 6, 7], [8, 8, 8]], dtype=torch.int32, device=dev())
    g = torch.Generator().manual_seed(3)
    p = sample_surface_points(v, f, 20000, generator=g, multinomial_max=0).cpu()
    assert float((p - torch.tensor([5.0, 5.0, 5.0])).abs().sum(dim=1).min()) > 1.0        # never the degenerate vertex
    assert bool(((p[:, 2].abs() < 1e-6) & (p[:, 1] >= 0) & (p[:, 1] <= 1)).all())
    far = float((p[:, 0] >= 10).float().mean())
    assert abs(far - 0.75) < 0.02, far                                 # 3 of 4 area units
    g2 = torch.Generator().manual_seed(3)
    assert torch.equal(p, sample_surface_points(v, f, 20000, generator=g2, multinomial_max=0).cpu())


def _closed_and_oriented(v, f):
    fl = f.long()
    e = torch.cat([fl[:, [0, 1]], fl[:, [1, 2]], fl[:, [2, 0]]]).sort(dim=1).values
    _, counts = torch.unique(e[:, 0] * v.shape[0] + e[:, 1], return_counts=True)
    de = torch.cat([fl[:, [0, 1]], fl[:, [1, 2]], fl[:, [2, 0]]])
    dkey = de[:, 0] * v.shape[0] + de[:, 1]
    closed = int(counts.min()) == 2 and int(counts.max()) == 2
    oriented = torch.unique(dkey).numel() == dkey.numel()
    return closed, oriented, v.shape[0] - counts.numel() + f.shape[0]


def test_marching_cubes_topology_on_analytic_shapes():
    """The derived 256-case table on shapes of known topology: a sphere (Euler characteristic 2) and a torus (0) come out
    closed, consistently oriented with outward normals (positive signed volume ~ the analytic volume), and with exactly
    the marching-cubes vertex count (one vertex per sign-changing grid edge) -- equal to the CPU statement's."""
    from ishapediting_amd.mesh import extract_surface
    res = 96
    for name, vol, chi, volume in (("sphere", sphere(res, 30.3), 2, 4 / 3 * np.pi * 30.3 ** 3),
                                   ("torus", torus(res, 28.0, 9.4), 0, 2 * np.pi ** 2 * 28.0 * 9.4 ** 2)):
        v, f = extract_surface(vol.to(dev()))
        closed, oriented, euler = _closed_and_oriented(v, f)
        assert closed and oriented and euler == chi, (name, closed, oriented, euler)
        assert v.shape[0] == S.mc_vertices(vol).shape[0]
        A, B, Cc = v[f[:, 0].long()].double(), v[f[:, 1].long()].double(), v[f[:, 2].long()].double()
        signed = float((A * torch.cross(B, Cc, dim=1)).sum() / 6)
        assert abs(signed - volume) < 0.01 * volume, (name, signed, volume)


def test_smoothing_of_an_open_surface_matches_unique_adjacency():
    """A sphere that leaves the volume is cut open by the box: edges in the box faces belong to one triangle.  The device
    smoothing (per-face gathering with boundary edges counted double, then halved) must equal Open3D's rule -- every
    neighbour once -- as stated in oracle/surface_cpu.py:smooth_simple."""
    from ishapediting_amd.mesh import extract_surface, smooth_mesh
    res = 32
    vol = sphere(res, 14.0, centre=8.0)                               # pokes through three faces of the box
    v, f = extract_surface(vol.to(dev()))
    closed, _, _ = _closed_and_oriented(v, f)
    assert not closed
    sm_d = smooth_mesh(v, f, 10, box_max=float(res - 1))
    sm_o = S.smooth_simple(v.cpu(), f.cpu().long(), 10)
    assert float((sm_d.cpu() - sm_o).abs().max()) < 1e-4
    wrong = smooth_mesh(v, f, 10)                                      # treating it as closed under-weights the rim
    assert float((wrong.cpu() - sm_o).abs().max()) > 1e-3


def test_mesh_chamfer_on_area_uniform_samples():
    """calc_chamfer (meshProcess.py:18-35) samples the SURFACES uniformly by area.  Two concentric spheres of radii r and
    r + d: every sample of one lies d from the other surface, so the distance is ~ 2 d^2, while vertex sampling would carry
    the grid's vertex density pattern."""
    from ishapediting_amd.mesh import extract_surface, mesh_chamfer
    res, r, d = 64, 20.0, 1.5
    a = extract_surface(sphere(res, r).to(dev()))
    b = extract_surface(sphere(res, r + d).to(dev()))
    c = mesh_chamfer(a, b, point_num=20000, seed=1)
    assert abs(c - 2 * d * d) < 0.08 * 2 * d * d, c
    # two independent samplings of ONE surface: the sampling floor, 2 / (pi * density) for uniform samples
    floor = 2 * (4 * np.pi * r * r) / (np.pi * 20000)
    assert abs(mesh_chamfer(a, a, point_num=20000, seed=1) - floor) < 0.15 * floor


def test_full_size_surface_is_a_closed_sphere():
    """256^3 (BASELINE's shape_resolution): an off-centre sphere SDF -> every edge lies in exactly two triangles,
    Euler characteristic 2, vertices on the sphere, the marching-cubes vertex count of the CPU statement, bitwise repeatable."""
    from ishapediting_amd.mesh import extract_surface
    r = 90.4
    vol = sphere(256, r, centre=120.3).to(dev())
    v, f = extract_surface(vol)
    v2, f2 = extract_surface(vol)
    assert torch.equal(v, v2) and torch.equal(f, f2)
    assert v.shape[0] == S.mc_vertices(vol.cpu()).shape[0]
    rad = torch.linalg.norm(v - 120.3, dim=1)
    assert float((rad - r).abs().max()) < 0.02
    fl = f.long()
    e = torch.cat([fl[:, [0, 1]], fl[:, [1, 2]], fl[:, [2, 0]]]).sort(dim=1).values
    key = e[:, 0] * v.shape[0] + e[:, 1]
    _, counts = torch.unique(key, return_counts=True)
    assert int(counts.min()) == 2 and int(counts.max()) == 2
    n_edges = counts.numel()
    assert v.shape[0] - n_edges + f.shape[0] == 2
    # consistent orientation: every directed edge appears exactly once
    de = torch.cat([fl[:, [0, 1]], fl[:, [1, 2]], fl[:, [2, 0]]])
    dkey = de[:, 0] * v.shape[0] + de[:, 1]
    assert torch.unique(dkey).numel() == dkey.numel()


def test_occupancy_mesh_and_obj_export(tmp_path):
    from ishapediting_amd.mesh import OccupancyMesh, export_obj, volume_to_mesh, write_mesh
    vol = sphere(32, 9.3).to(dev())
    m = volume_to_mesh(vol, 32, smooth_iterations=10)
    assert isinstance(m, OccupancyMesh)                               # the backend is explicit ("device"), never an import probe
    with pytest.raises(ValueError):
        volume_to_mesh(vol, 32, backend="whatever")
    if True:
        nv, nt = m.counts()
        assert m.vertices.shape == (nv, 3) and m.triangles.shape == (nt, 3)
        assert float(m.vertices.abs().max()) <= 1.0                   # visualize.py:101 convention: grid / res * 2 - 1
        write_mesh(str(tmp_path / "m.obj"), m)
        lines = open(tmp_path / "m.obj").read().splitlines()
        assert sum(l.startswith("v ") for l in lines) == nv and sum(l.startswith("f ") for l in lines) == nt
    export_obj(vol, str(tmp_path / "e.obj"), scale_div=31.0)
    assert open(tmp_path / "e.obj").read().startswith("v ")


def test_mesh_occupancy_and_surface_sampling(tmp_path):
    """train_triplane's data preparation without Open3D: a closed mesh (the sphere surface extracted above, scaled to
    [-1,1]) sampled on the device.  Occupancy vs the CPU statement (same rule; <= 0.05 % of the points may differ, those
    whose ray grazes an edge within fp32 rounding) and vs the analytic sphere away from the surface; surface samples lie
    on the mesh; the file route (OBJ) gives the same occupancy."""
    from ishapediting_amd.mesh import extract_surface, mesh_occupancy, sample_occupancy, sample_surface_points, _write_obj
    res, r = 48, 15.2
    v, f = extract_surface(sphere(res, r).to(dev()))
    v = v / (res - 1) * 2 - 1                                      # grid -> [-1, 1]
    rad = r / (res - 1) * 2
    g = torch.Generator().manual_seed(2)
    pts = (torch.rand((20000, 3), generator=g) * 2 - 1).to(dev())
    occ = mesh_occupancy(v, f, pts)
    ref = S.mesh_occupancy(v.cpu(), f.cpu(), pts.cpu())
    assert float((occ.cpu() != ref).float().mean()) <= 5e-4
    d = torch.linalg.norm(pts, dim=1)
    clear = (d - rad).abs() > 0.03                                 # the mesh is a polyhedral approximation of the sphere
    assert torch.equal(occ[clear], (d[clear] < rad).float())
    assert 0.05 < float(occ.mean()) < 0.5                         # sphere volume / cube volume = 0.135 here
    sp = sample_surface_points(v, f, 5000, generator=g)
    assert float((torch.linalg.norm(sp, dim=1) - rad).abs().max()) < 0.02
    assert float(sp.mean(dim=0).abs().max()) < 0.03               # area-weighted: centred on the sphere
    _write_obj(str(tmp_path / "s.obj"), v, f)
    p2, o2 = sample_occupancy(None, str(tmp_path / "s.obj"), True, 4000, 0.5, device=dev(), generator=g)
    assert p2.shape == (4000, 3) and o2.shape == (4000, 1) and set(np.unique(o2)) <= {0.0, 1.0}
    d2 = np.linalg.norm(p2, axis=1)
    far = np.abs(d2 - rad) > 0.03
    np.testing.assert_array_equal(o2[far, 0], (d2[far] < rad).astype(np.float32))
