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
