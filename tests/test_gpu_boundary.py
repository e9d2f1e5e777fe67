"""Round-3 boundary tests: the triplane_decoder/visualize.py call surface and the mesh hand-off
to the reference's Open3D GUI.  Open3D / PyMCubes are absent in this image, so the Open3D side is a stub module with
the three classes the hand-off touches (geometry.TriangleMesh, utility.Vector3dVector / Vector3iVector); everything else
is the real device path.  Needs an MI355X: -m gpu."""
import copy
import os
import sys
import types
from argparse import Namespace

import numpy as np
import pytest
import torch

from ishapediting_amd import mesh as mesh_backend
from ishapediting_amd import synthetic, visualize
from ishapediting_amd.triplane_decoder import MultiTriplane

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


class _StubMesh:
    """The attributes main.py reads / writes on an open3d.geometry.TriangleMesh (main.py:373-379,507)."""

    def __init__(self):
        self.vertices, self.triangles, self.vertex_normals = np.zeros((0, 3)), np.zeros((0, 3), np.int32), np.zeros((0, 3))

    def has_vertex_normals(self):
        return len(self.vertex_normals) > 0

    def has_triangle_normals(self):
        return False


@pytest.fixture
def stub_open3d(monkeypatch):
    o3d = types.ModuleType("open3d")
    o3d.geometry = types.SimpleNamespace(TriangleMesh=_StubMesh)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a, np.float64),
                                        Vector3iVector=lambda a: np.asarray(a, np.int32))
    monkeypatch.setitem(sys.modules, "open3d", o3d)
    return o3d


def sphere_decoder(S=32):
    """A decoder + planes whose level set is a closed surface: smooth low-amplitude planes make the random-weight MLP a
    smooth function of position (tools/parity_report.py's shape-like case)."""
    dec = MultiTriplane(1, device=dev())
    dec.net.load_state_dict(synthetic.decoder_state_dict())
    g = torch.Generator().manual_seed(5)
    ax = torch.linspace(-1, 1, S)
    yy, xx = torch.meshgrid(ax, ax, indexing="ij")
    base = torch.stack([torch.cos(1.5 * xx + k) * torch.sin(1.1 * yy - k) for k in range(32)]) * 0.05
    for i in range(3):
        dec.embeddings[i] = (base + 0.01 * torch.randn(32, S, S, generator=g)).unsqueeze(0).to(dev())
    return dec


def test_visualize_create_obj_and_main_write_the_reference_files(tmp_path):
    """create_obj(model, obj_idx, res, max_batch_size, output_path) and main(args, feature) (visualize.py:36-73,108-128):
    an OBJ whose vertices are grid / 255 * 2 - 1 of the level-0 surface of the decoded grid."""
    dec = sphere_decoder()
    res = 48
    vol = visualize.decode_grid(dec, 0, res)
    vol = vol - vol.median()                              # put the level set inside the volume
    # through the function itself (its volume is returned for inspection)
    out = tmp_path / "a.obj"
    v2 = visualize.create_obj(dec, 0, res=res, max_batch_size=50000, output_path=str(out))
    assert v2.shape == (res, res, res) and out.exists()
    nv, nt = mesh_backend.surface_counts(v2, 0.0)
    verts, tris = mesh_backend.read_obj(str(out))
    assert verts.shape[0] == nv and tris.shape[0] == nt
    if nv:
        gv, _ = mesh_backend.extract_surface(v2, 0.0)
        np.testing.assert_allclose(verts, (gv / 255.0 * 2 - 1).cpu().numpy(), atol=2e-6)
    # main(): triplanes from `feature`, weights from state_dict (no checkpoint files offline), same bytes as create_obj
    feat = np.concatenate([dec.embeddings[i].cpu().numpy() for i in range(3)], axis=0)      # [3,32,S,S]
    out2 = tmp_path / "b.obj"
    visualize.main(Namespace(input=None, output=str(out2), model_path=None, res=res), feature=feat,
                   state_dict=synthetic.decoder_state_dict())
    assert out2.read_bytes() == out.read_bytes()
    # and from a .npy file, the generate.py route (generate.py:88-95)
    np.save(tmp_path / "t.npy", feat.reshape(96, 32, 32))
    out3 = tmp_path / "c.obj"
    visualize.main(Namespace(input=str(tmp_path / "t.npy"), output=str(out3), model_path=None, res=res),
                   state_dict=synthetic.decoder_state_dict())
    assert out3.read_bytes() == out.read_bytes()


def test_create_obj_o3d_hands_an_open3d_mesh_to_the_gui(stub_open3d, monkeypatch):
    """create_obj_o3d(model, obj_idx, res, max_batch_size) (visualize.py:76-105) with mesh.BACKEND = "open3d": an
    open3d.geometry.TriangleMesh (the stub's class here) carrying the device surface, vertices / res * 2 - 1, unsmoothed;
    and with the default backend an OccupancyMesh with the TriangleMesh methods the GUI calls (main.py:373,507)."""
    dec = sphere_decoder()
    res = 48
    m_dev = visualize.create_obj_o3d(dec, 0, res=res, max_batch_size=50000)
    assert isinstance(m_dev, mesh_backend.OccupancyMesh) and m_dev.smooth_iterations == 0
    assert not m_dev.has_vertex_normals() and not m_dev.has_triangle_normals()
    v = m_dev.vertices_numpy()
    assert v.dtype == np.float64 and v.shape[1] == 3 and m_dev.triangles_numpy().dtype == np.int32
    monkeypatch.setattr(mesh_backend, "BACKEND", "open3d")
    m = visualize.create_obj_o3d(dec, 0, res=res)
    assert isinstance(m, _StubMesh)
    np.testing.assert_array_equal(np.asarray(m.vertices), v)
    np.testing.assert_array_equal(np.asarray(m.triangles), m_dev.triangles_numpy())
    # normals: unit length, pointing along the triangle normals' area-weighted sum; deepcopy keeps them
    if v.shape[0]:
        m_dev.compute_vertex_normals()
        assert m_dev.has_vertex_normals()
        n = np.asarray(m_dev.vertex_normals)
        assert n.shape == v.shape and np.allclose(np.linalg.norm(n, axis=1)[np.isfinite(n).all(1)], 1.0, atol=1e-4)
        m2 = copy.deepcopy(m_dev)
        assert m2.has_vertex_normals() and np.array_equal(np.asarray(m2.vertex_normals), n)
        assert len(m_dev.to_open3d().vertex_normals) == v.shape[0]


def test_train_triplane_accepts_the_guis_mesh_object(stub_open3d):
    """main.py:447-451 calls train_triplane(mesh=<open3d TriangleMesh>): on the device route any object with `.vertices` /
    `.triangles` is sampled (mesh.sample_occupancy), not only files and tuples -- round 2 fell into read_obj(None)."""
    # a unit octahedron as the "GUI mesh"
    m = _StubMesh()
    m.vertices = np.array([[0.6, 0, 0], [-0.6, 0, 0], [0, 0.6, 0], [0, -0.6, 0], [0, 0, 0.6], [0, 0, -0.6]], np.float64)
    m.triangles = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int32)
    g = torch.Generator().manual_seed(0)
    pts, occ = mesh_backend.sample_occupancy(m, None, True, 4000, 0.5, device=dev(), generator=g)
    assert pts.shape == (4000, 3) and occ.shape == (4000, 1)
    inside = (np.abs(pts).sum(1) < 0.6)
    near = np.abs(np.abs(pts).sum(1) - 0.6) < 1e-3
    assert ((occ[:, 0] > 0.5) == inside)[~near].all()                 # exact octahedron: |x|+|y|+|z| < 0.6
    # the tuple and OccupancyMesh forms still work
    v, t = mesh_backend.mesh_arrays((m.vertices, m.triangles))
    assert v.dtype == np.float32 and t.dtype == np.int32 and v.shape == (6, 3) and t.shape == (8, 3)
