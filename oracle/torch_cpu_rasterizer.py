"""Pure-PyTorch, autograd-differentiable CPU restatement of the rasterizer -- TEST INFRASTRUCTURE.

Independent check of the hand-derived backward (DGR/cuda_rasterizer/backward.cu) and the "PyTorch-CPU
autograd rasterize" numerics reference of BASELINE config 1.  The discrete parts (tile lists, their
depth order) are taken from the C oracle (oracle/cpu_rasterizer.c); everything differentiable is
recomputed here from the raw inputs with torch ops, so autograd yields d(image)/d(inputs) without
using any of the reference's analytic gradient formulas.

Reference semantics reproduced on purpose (they are not "mathematically clean", but they are what
DGR computes -- SURVEY.md appendix A.5/A.6):
  * no gradient through the power>0 / alpha<1/255 / T<1e-4 tests (piecewise constant);
  * min(0.99, .) is straight-through in the backward (backward.cu:499-554 applies the unclamped formula);
  * when t.x/t.z is clamped to 1.3*tanfov the clamped value is treated as a constant
    (x_grad_mul, backward.cu:175-176,262-264);
  * means2D gradient is reported in NDC-scaled units (x 0.5*W, 0.5*H; backward.cu:460-461,545-546).
"""
from __future__ import annotations

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh_color(deg, shs, dirs):
    """DGR/cuda_rasterizer/forward.cu:20-71; shs [P,M,3], dirs [P,3] normalised."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = SH_C0 * shs[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            r = (r + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
                 + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                r = (r + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
                     + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                     + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                     + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(r + 0.5, 0.0)


def quat_to_rotmat(q):
    """Standard rotation matrix of a (real-first, un-normalised as given) quaternion; forward.cu:127-138."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


def cov3d_from_scale_rot(scales, mod, rots):
    R = quat_to_rotmat(rots)
    S = (mod * scales)
    RS = R * S[:, None, :]
    return RS @ RS.transpose(1, 2)  # R S^2 R^T


def unpack_cov6(c6):
    P = c6.shape[0]
    S = c6.new_zeros(P, 3, 3)
    S[:, 0, 0] = c6[:, 0]; S[:, 0, 1] = c6[:, 1]; S[:, 0, 2] = c6[:, 2]
    S[:, 1, 0] = c6[:, 1]; S[:, 1, 1] = c6[:, 3]; S[:, 1, 2] = c6[:, 4]
    S[:, 2, 0] = c6[:, 2]; S[:, 2, 1] = c6[:, 4]; S[:, 2, 2] = c6[:, 5]
    return S


def render(means3D, means2D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
           cov3D_precomp=None, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=3,
           scale_modifier=1.0, point_list, ranges, radii):
    """Differentiable image [3,H,W].  point_list / ranges / radii come from the C oracle."""
    dt = means3D.dtype
    V = viewmatrix.to(dt); PM = projmatrix.to(dt); campos = campos.to(dt); bg = bg.to(dt)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], dim=1)
    p_view = hom @ V[:, :3]
    p_hom = hom @ PM
    p_w = 1.0 / (p_hom[:, 3:4] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w
    WH = torch.tensor([W, H], dtype=dt)
    pix = ((p_proj[:, :2] + means2D[:, :2] + 1.0) * WH - 1.0) * 0.5

    Sigma = unpack_cov6(cov3D_precomp) if cov3D_precomp is not None else cov3d_from_scale_rot(scales, scale_modifier, rotations)
    fx = W / (2.0 * tanfovx); fy = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    in_x = (txtz >= -limx) & (txtz <= limx)
    in_y = (tytz >= -limy) & (tytz <= limy)
    tx = torch.where(in_x, p_view[:, 0], (txtz.clamp(-limx, limx) * tz).detach())
    ty = torch.where(in_y, p_view[:, 1], (tytz.clamp(-limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).reshape(P, 2, 3)
    Rw2c = V[:3, :3].transpose(0, 1)  # p_view = Rw2c @ p + t
    A = J @ Rw2c
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3; b = cov2[:, 0, 1]; c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], dim=1)

    if colors_precomp is not None:
        col = colors_precomp
    else:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        col = eval_sh_color(sh_degree, shs, d)

    op = opacities.reshape(-1)
    out = torch.zeros(3, H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    gx = (W + 15) // 16
    pl = torch.as_tensor(point_list.astype("int64"))
    for tile in range(ranges.shape[0]):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        tx0, ty0 = (tile % gx) * 16, (tile // gx) * 16
        xs = torch.arange(tx0, min(tx0 + 16, W)); ys = torch.arange(ty0, min(ty0 + 16, H))
        if len(xs) == 0 or len(ys) == 0:
            continue
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        pxf = xx.reshape(-1).to(dt); pyf = yy.reshape(-1).to(dt)
        npx = pxf.shape[0]
        if r1 == r0:
            out[:, yy.reshape(-1), xx.reshape(-1)] = bg[:, None].expand(3, npx)
            continue
        ids = pl[r0:r1]
        dx = pix[ids, 0][None, :] - pxf[:, None]
        dy = pix[ids, 1][None, :] - pyf[:, None]
        cn = conic[ids]
        power = -0.5 * (cn[:, 0][None] * dx * dx + cn[:, 2][None] * dy * dy) - cn[:, 1][None] * dx * dy
        raw = op[ids][None] * torch.exp(power.clamp_max(0.0))
        alpha = raw + (torch.clamp_max(raw, 0.99) - raw).detach()  # straight-through min(0.99, .)
        valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        with torch.no_grad():
            one_m = 1.0 - a_eff
            T_incl = torch.cumprod(one_m, dim=1)  # test_T after each entry
            stop = valid & (T_incl < 0.0001)
            n = a_eff.shape[1]
            idx = torch.arange(n)[None].expand_as(stop)
            first_stop = torch.where(stop, idx, torch.full_like(idx, n)).min(dim=1).values
            keep = idx < first_stop[:, None]
            contrib_pos = torch.where(valid & keep, idx + 1, torch.zeros_like(idx)).max(dim=1).values
        a_eff = torch.where(keep, a_eff, torch.zeros_like(a_eff))
        one_m = 1.0 - a_eff
        T_incl = torch.cumprod(one_m, dim=1)
        T_excl = torch.cat([torch.ones(npx, 1, dtype=dt), T_incl[:, :-1]], dim=1)
        wgt = a_eff * T_excl
        Cc = wgt @ col[ids]  # [npx,3]
        T_fin = T_incl[:, -1]
        res = Cc + T_fin[:, None] * bg[None]
        out[:, yy.reshape(-1), xx.reshape(-1)] = res.transpose(0, 1)
        n_contrib[yy.reshape(-1), xx.reshape(-1)] = contrib_pos
    return out, n_contrib
