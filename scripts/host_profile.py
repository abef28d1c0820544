"""host-side cost of one forward + backward through the Python API (tiny scene: the kernels are negligible)"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn
from sugar_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
scene, cams, bg = syn.make_config("config1")
cam = cams[0]
st = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, bg.to(dev), 1.0, cam.viewmatrix.to(dev),
                                   cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
rast = GaussianRasterizer(st)
leaves = [t.to(dev).requires_grad_(True) for t in (scene.means3D, scene.opacities, scene.shs, scene.scales, scene.rotations)]
m2 = torch.zeros_like(leaves[0], requires_grad=True)
g = torch.randn(3, cam.image_height, cam.image_width, device=dev)
def step():
    color, radii = rast(leaves[0], m2, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    color.backward(g)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("fwd+bwd wall per call: %.1f us" % (1e6 * (time.perf_counter() - t0) / 200))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
