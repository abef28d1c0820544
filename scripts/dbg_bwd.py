import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sugar_amd import _lib, synthetic as syn
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
lib = _lib.load(); dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = 64, 48
scene = syn.make_scene(P, 3, 0.02, 0.15)
cam = syn.orbit_cameras(W, H)[1]
g = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
accs = []
for v in (1, 3):
    lib.sgr_set_blend_variant(v)
    st = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, torch.tensor([0.1, 0.2, 0.3]).to(dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
    leaves = [t.to(dev).requires_grad_(True) for t in (scene.means3D, scene.opacities, scene.shs, scene.scales, scene.rotations)]
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii = GaussianRasterizer(st)(means3D=leaves[0], means2D=m2, opacities=leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    color.backward(g)
    torch.cuda.synchronize()
    geom = _C.last_forward["geom"]
    off = (P * 48 + 255) // 256 * 256
    accs.append(geom[off: off + P * 48].view(torch.float32).reshape(P, 12).cpu().numpy().copy())
    print("variant", v, "R", _C.last_forward["num_rendered"], "acc finite", np.isfinite(accs[-1]).all(), "absmax", np.abs(accs[-1]).max())
a, b = accs
d = np.abs(a - b).max(axis=1)
bad = np.argsort(-d)[:8]
np.set_printoptions(precision=4, suppress=False, linewidth=200)
print("n differing rows:", (d > 1e-3 * (np.abs(a).max(axis=1) + 1e-6)).sum(), "of", P)
for i in bad:
    print(i, "\n  old", a[i, :9], "\n  new", b[i, :9])
