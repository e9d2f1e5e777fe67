"""Timing of ishap_mesh_smooth (10 sweeps) on a 256^3 noise volume and on a sphere; prints ms and a checksum."""
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ishapediting_amd.mesh import extract_surface, smooth_mesh
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
RES = 256
noise = torch.randn((RES,) * 3, generator=g).to(dev)
ax = torch.arange(RES, dtype=torch.float32, device=dev) - 120.3
sph = 90.4 - torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
for name, vol in (("noise", noise), ("sphere", sph)):
    v, f = extract_surface(vol)
    smooth_mesh(v, f, 10, box_max=float(RES - 1))
    torch.cuda.synchronize()
    t0 = time.time()
    out = smooth_mesh(v, f, 10, box_max=float(RES - 1))
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 1e3
    out2 = smooth_mesh(v, f, 10, box_max=float(RES - 1))
    print(f"{name}: {v.shape[0]} vertices, {f.shape[0]} triangles, 10 sweeps {ms:.2f} ms, bitwise repeatable {bool(torch.equal(out, out2))}, "
          f"checksum {out.double().sum().item():.6f}")
