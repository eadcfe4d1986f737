"""Oracle (test infrastructure): pixel -> camera -> SE3 -> pixel chain and bilinear sampling.

Restates reference ``pytorch_version/inverse_warp.py`` (duplicated verbatim in
``loss_functions.py:44-231``).  Closed form per target pixel (u, v):

    cam   = (Kinv @ [u, v, 1]) * depth                      inverse_warp.py:26-40
    [R|t] = pose_vec2mat(pose)                              inverse_warp.py:141-157
    P     = K @ [R|t]                                       inverse_warp.py:188
    p     = P[:, :3] @ cam + P[:, 3]                        inverse_warp.py:55-60
    Z     = max(p_z, 1e-3)                                  inverse_warp.py:63
    x_n   = 2 (p_x / Z) / (W - 1) - 1 ; y_n likewise        inverse_warp.py:65-66
    zeros padding: x_n or y_n outside [-1, 1] -> 2          inverse_warp.py:67-71
    sample: bilinear, 4 taps, OOB taps = 0                  inverse_warp.py:191

``F.grid_sample`` is called by the reference without ``align_corners``; under
the installed torch that means ``align_corners=False`` (SURVEY.md preamble #5),
so un-normalisation is ``ix = ((x_n + 1) * W - 1) / 2``.  That choice is frozen
here as the default; ``align_corners=True`` is the opt-in.
"""
import torch


def pixel_grid(h, w, dtype, device="cpu"):
    """(u, v, 1) per pixel, [1, 3, H, W].  inverse_warp.py:8-15 (set_id_grid)."""
    v = torch.arange(0, h, dtype=dtype, device=device).view(1, h, 1).expand(1, h, w)
    u = torch.arange(0, w, dtype=dtype, device=device).view(1, 1, w).expand(1, h, w)
    return torch.stack((u, v, torch.ones(1, h, w, dtype=dtype, device=device)), dim=1)


def pixel2cam(depth, intrinsics_inv):
    """depth [B,H,W], Kinv [B,3,3] -> cam [B,3,H,W].  inverse_warp.py:26-40."""
    b, h, w = depth.shape
    grid = pixel_grid(h, w, depth.dtype, depth.device).expand(b, 3, h, w).reshape(b, 3, -1)
    cam = (intrinsics_inv @ grid).reshape(b, 3, h, w)
    return cam * depth.unsqueeze(1)


def euler2mat(angle):
    """[B,3] (x, y, z) -> R = Rx @ Ry @ Rz, [B,3,3].  inverse_warp.py:77-114."""
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    cz, sz = torch.cos(z), torch.sin(z)
    cy, sy = torch.cos(y), torch.sin(y)
    cx, sx = torch.cos(x), torch.sin(x)
    o = torch.zeros_like(z)
    i = torch.ones_like(z)
    rz = torch.stack([cz, -sz, o, sz, cz, o, o, o, i], dim=1).reshape(-1, 3, 3)
    ry = torch.stack([cy, o, sy, o, i, o, -sy, o, cy], dim=1).reshape(-1, 3, 3)
    rx = torch.stack([i, o, o, o, cx, -sx, o, sx, cx], dim=1).reshape(-1, 3, 3)
    return rx @ ry @ rz


def quat2mat(quat):
    """[B,3] = last three coefficients of a (1, q) quaternion, normalised.  inverse_warp.py:117-138."""
    q = torch.cat([torch.ones_like(quat[:, :1]), quat], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).reshape(-1, 3, 3)


def pose_vec2mat(vec, rotation_mode="euler"):
    """[B,6] = (tx,ty,tz,rx,ry,rz) -> [B,3,4].  inverse_warp.py:141-157."""
    t = vec[:, :3].unsqueeze(-1)
    r = euler2mat(vec[:, 3:]) if rotation_mode == "euler" else quat2mat(vec[:, 3:])
    return torch.cat([r, t], dim=2)


def cam2pixel(cam, proj_rot, proj_tr, padding_mode):
    """cam [B,3,H,W] -> normalised sampling grid [B,H,W,2].  inverse_warp.py:43-74."""
    b, _, h, w = cam.shape
    p = proj_rot @ cam.reshape(b, 3, -1) + proj_tr
    X, Y = p[:, 0], p[:, 1]
    Z = p[:, 2].clamp(min=1e-3)
    xn = 2 * (X / Z) / (w - 1) - 1
    yn = 2 * (Y / Z) / (h - 1) - 1
    if padding_mode == "zeros":
        # non-differentiable overwrite (index_put in the reference): gradient is cut there
        xn = torch.where(((xn > 1) | (xn < -1)).detach(), torch.full_like(xn, 2.0), xn)
        yn = torch.where(((yn > 1) | (yn < -1)).detach(), torch.full_like(yn, 2.0), yn)
    return torch.stack([xn, yn], dim=2).reshape(b, h, w, 2)


def bilinear_sample(img, grid, padding_mode="zeros", align_corners=False):
    """Explicit restatement of aten ``grid_sampler_2d`` (bilinear) as called at
    inverse_warp.py:191: un-normalise, 4 taps nw/ne/sw/se, out-of-bounds taps
    contribute 0 (zeros) or coordinates are clipped to the border first (border)."""
    b, c, h, w = img.shape
    x, y = grid[..., 0], grid[..., 1]
    if align_corners:
        ix = (x + 1) / 2 * (w - 1)
        iy = (y + 1) / 2 * (h - 1)
    else:
        ix = ((x + 1) * w - 1) / 2
        iy = ((y + 1) * h - 1) / 2
    if padding_mode == "border":
        ix = ix.clamp(0, w - 1)
        iy = iy.clamp(0, h - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    wnw = (x1 - ix) * (y1 - iy)
    wne = (ix - x0) * (y1 - iy)
    wsw = (x1 - ix) * (iy - y0)
    wse = (ix - x0) * (iy - y0)
    flat = img.reshape(b, c, h * w)

    def tap(xi, yi, wt):
        ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
        idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long().reshape(b, 1, -1).expand(b, c, -1)
        val = torch.gather(flat, 2, idx).reshape(b, c, *xi.shape[1:])
        return val * (wt * ok.to(img.dtype)).unsqueeze(1)

    # accumulation order nw, ne, sw, se as in aten (matters for the exact-zero OOB test)
    return tap(x0, y0, wnw) + tap(x1, y0, wne) + tap(x0, y1, wsw) + tap(x1, y1, wse)


def inverse_warp(img, depth, pose, intrinsics, intrinsics_inv, rotation_mode="euler",
                 padding_mode="zeros", align_corners=False):
    """Source image sampled at the reprojection of every target pixel.  inverse_warp.py:160-193."""
    assert depth.dim() == 3 and pose.dim() == 2 and pose.shape[1] == 6
    assert intrinsics.shape == intrinsics_inv.shape
    cam = pixel2cam(depth, intrinsics_inv)
    proj = intrinsics @ pose_vec2mat(pose, rotation_mode)
    grid = cam2pixel(cam, proj[:, :, :3], proj[:, :, -1:], padding_mode)
    return bilinear_sample(img, grid, padding_mode, align_corners)


# ------------------------------------------------------------------------------------------------------------------
# unsupervise_dvo.py front end (Caffe-style chain).  The Python port in the reference (geo_transform.py) is dead code
# (exit(0) at :31), so these restate the Caffe layers it was ported from; only the SE3 exponential map has a runnable
# Python twin (se3_generate.py) and golden vectors (tests/golden/se3_expmap.npz).  The pixel-coordinate sampling
# itself is "parity unpinned": it is cross-checked against the align_corners=True grid path instead.

class _Se3Exp(torch.autograd.Function):
    """Forward AND hand-written backward of the reference's SE3_Generator_KITTI (se3_generate.py:7-103), restated
    with batched torch ops.  The backward is the reference's, not autograd's: for th^2 < 1e-12 it uses the constant
    "generators" exactly as the reference writes them (se3_generate.py:75-78), whose sign is the opposite of
    d(I + [w]x)/dw -- a quirk of the reference that the parity target keeps."""

    @staticmethod
    def forward(ctx, vec):
        w, u = vec[:, :3], vec[:, 3:]
        z = torch.zeros_like(w[:, 0])
        wx = torch.stack([z, -w[:, 2], w[:, 1], w[:, 2], z, -w[:, 0], -w[:, 1], w[:, 0], z], dim=1).reshape(-1, 3, 3)
        th2 = (w * w).sum(1)
        small = th2 < 1e-12                                                # se3_generate.py:11,33
        th = torch.sqrt(torch.where(small, torch.ones_like(th2), th2))
        c1 = torch.where(small, torch.ones_like(th), torch.sin(th) / th)   # :37
        c2 = torch.where(small, torch.zeros_like(th), 2 * torch.sin(th / 2) ** 2 / th ** 2)   # :38
        eye = torch.eye(3, dtype=vec.dtype).expand_as(wx)
        R = eye + c1.view(-1, 1, 1) * wx + c2.view(-1, 1, 1) * (wx @ wx)    # :34,42
        ctx.save_for_backward(w, u, wx, R, th2)
        return torch.cat([R, R @ u.unsqueeze(-1)], dim=2)                  # :45-47

    @staticmethod
    def backward(ctx, g):
        w, u, wx, R, th2 = ctx.saved_tensors
        gT, gR = g[:, :, 3], g[:, :, :3]
        g_u = (gT.unsqueeze(1) @ R).squeeze(1)                              # dLdut = dLdT x R, :64
        gR = gR + gT.unsqueeze(2) * u.unsqueeze(1)                          # grad_corr, :67-71
        gens = torch.tensor([[[0, 0, 0], [0, 0, 1], [0, -1, 0]], [[0, 0, -1], [0, 0, 0], [1, 0, 0]],
                             [[0, 1, 0], [-1, 0, 0], [0, 0, 0]]], dtype=g.dtype)   # :75-78
        eye = torch.eye(3, dtype=g.dtype)
        small = th2 < 1e-12
        g_w = []
        for i in range(3):
            cross_term = (wx @ (eye - R)[:, :, i:i + 1]).squeeze(-1)       # uw_x (I - R) e_i, :85
            z = torch.zeros_like(cross_term[:, 0])
            cross = torch.stack([z, -cross_term[:, 2], cross_term[:, 1], cross_term[:, 2], z, -cross_term[:, 0],
                                 -cross_term[:, 1], cross_term[:, 0], z], dim=1).reshape(-1, 3, 3)   # :86-92
            safe = torch.where(small, torch.ones_like(th2), th2)
            dR = ((w[:, i].view(-1, 1, 1) * wx + cross) / safe.view(-1, 1, 1)) @ R                 # :98
            dR = torch.where(small.view(-1, 1, 1), gens[i].expand_as(dR), dR)                      # :96
            g_w.append((gR * dR).sum((1, 2)))                                                      # :99
        return torch.cat([torch.stack(g_w, dim=1), g_u], dim=1)


def se3_exp(vec):
    """(wx,wy,wz,ux,uy,uz) [B,6] -> [R | R u] [B,3,4] with the reference's own backward (see _Se3Exp)."""
    return _Se3Exp.apply(vec)


def inverse_warp_pixel(img, depth, T, K):
    """Caffe GeoTransform -> PinHole -> InverseWarping (geometry_transformation.cu:10-47, pin_hole_layer.cu:10-50,
    inverse_warping_layer.cu:10-52): X = (x-cx)/fx d, Y = (y-cy)/fy d; p = T [X,Y,d,1]; u = fx px/(pz+1e-12) + cx,
    v likewise; bilinear gather at (u, v) in PIXEL units, each of the 4 taps bounds-checked (out-of-bounds taps add 0).
    img [B,C,H,W], depth [B,H,W], T [B,3,4], K [B,3,3] (fx, fy, cx, cy read from it)."""
    b, c, h, w = img.shape
    fx, fy, cx, cy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=img.dtype), torch.arange(w, dtype=img.dtype), indexing="ij")
    X = (xs[None] - cx.view(-1, 1, 1)) / fx.view(-1, 1, 1) * depth
    Y = (ys[None] - cy.view(-1, 1, 1)) / fy.view(-1, 1, 1) * depth
    pts = torch.stack([X, Y, depth, torch.ones_like(depth)], dim=1).reshape(b, 4, -1)
    p = (T @ pts).reshape(b, 3, h, w)
    u = fx.view(-1, 1, 1) * p[:, 0] / (p[:, 2] + 1e-12) + cx.view(-1, 1, 1)
    v = fy.view(-1, 1, 1) * p[:, 1] / (p[:, 2] + 1e-12) + cy.view(-1, 1, 1)
    x0, y0 = torch.floor(u), torch.floor(v)
    flat = img.reshape(b, c, h * w)
    out = 0
    for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):                 # nw, ne, sw, se
        xi, yi = x0 + dx, y0 + dy
        wt = (1 - (u - xi).abs()) * (1 - (v - yi).abs())
        ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
        idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long().reshape(b, 1, -1).expand(b, c, -1)
        out = out + torch.gather(flat, 2, idx).reshape(b, c, h, w) * (wt * ok.to(img.dtype)).unsqueeze(1)
    return out


def dvo_photometric_loss(img_R2, img_L2, img_R1, depth, T_R2L, T_2to1, K, caffe_abs=False):
    """LR_error + R12_error of unsupervise_dvo.py:96-117: se3 exp-map poses, pixel-coordinate warps of the left image
    (stereo pose) and of the previous right image (temporal pose), exact-zero mask, L1 mean each.
    caffe_abs: the Caffe graph's form instead (train.prototxt:4428-4446): AbsLoss(warped, target) -- per-sample sum,
    no validity mask, AbsLoss's sign convention."""
    from .losses import abs_loss_caffe
    loss = 0
    for src, pose in ((img_L2, T_R2L), (img_R1, T_2to1)):
        warped = inverse_warp_pixel(src, depth, se3_exp(pose), K)
        if caffe_abs:
            loss = loss + abs_loss_caffe(warped, img_R2)
            continue
        valid = 1 - (warped == 0).prod(1, keepdim=True).type_as(warped)
        loss = loss + ((img_R2 - warped) * valid).abs().mean()
    return loss
