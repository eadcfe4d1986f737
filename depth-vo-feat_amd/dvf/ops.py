"""torch.autograd.Function wrappers over the C ABI: geometry + losses.

Each forward/backward is one (or two) kernel launches on torch's current stream; nothing here
synchronises with the host, so a whole training step can be captured in a HIP graph.
"""
import torch

from . import lib as L


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class InverseWarpFn(torch.autograd.Function):
    """inverse_warp.inverse_warp (reference pytorch_version/inverse_warp.py:160-193)."""

    @staticmethod
    def forward(ctx, img, depth, pose, K, Kinv, flags):
        img, depth, pose, K, Kinv = map(_f32c, (img, depth, pose, K, Kinv))
        B, C, H, W = img.shape
        out = torch.empty_like(img)
        L.check(L.lib().dvf_inverse_warp_fwd(L.dev(img, "img"), L.dev(depth, "depth"), L.dev(pose, "pose"),
                                             L.dev(K, "intrinsics"), L.dev(Kinv, "intrinsics_inv"), L.dev(out),
                                             B, C, H, W, flags, L.stream()), "dvf_inverse_warp_fwd")
        ctx.save_for_backward(img, depth, pose, K, Kinv)
        ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, gout):
        img, depth, pose, K, Kinv = ctx.saved_tensors
        B, C, H, W = img.shape
        need_img, need_depth, need_pose = ctx.needs_input_grad[:3]
        g_img = torch.zeros_like(img) if need_img else None
        g_depth = torch.empty_like(depth) if need_depth else None
        g_pose = torch.empty_like(pose) if need_pose else None
        ws = torch.empty(int(L.lib().dvf_pose_ws_floats(1, B)), device=img.device) if need_pose else None
        L.check(L.lib().dvf_inverse_warp_bwd(L.dev(img), L.dev(depth), L.dev(pose), L.dev(K), L.dev(Kinv),
                                             L.dev(_f32c(gout), "grad_out"), L.dev(g_img), L.dev(g_depth),
                                             L.dev(g_pose), L.dev(ws), B, C, H, W, ctx.flags, L.stream()),
                "dvf_inverse_warp_bwd")
        return g_img, g_depth, g_pose, None, None, None


class PhotoLossFn(torch.autograd.Function):
    """Fused warp + exact-zero mask + (explainability mask) + L1 mean for all views of one scale.

    Inputs: tgt [B,C,H,W], depth [B,H,W], pose [V,B,6], K, Kinv [B,3,3], mask [B,V,H,W] or None,
    flags (int, or (int, in_scale): every image value is used as in_scale * x, the `0.004 * img` of
    unsupervise.py:101 without materialising scaled copies), then the V source tensors.
    Output: 0-dim loss (sum over views)."""

    @staticmethod
    def forward(ctx, tgt, depth, pose, K, Kinv, mask, flags, *srcs):
        flags, in_scale = flags if isinstance(flags, tuple) else (flags, 1.0)
        tgt, depth, pose, K, Kinv = map(_f32c, (tgt, depth, pose, K, Kinv))
        srcs = [_f32c(s) for s in srcs]
        mask = _f32c(mask) if mask is not None else None
        B, C, H, W = tgt.shape
        V = len(srcs)
        lib = L.lib()
        out = torch.empty(1 + V, device=tgt.device)
        partials = torch.empty(int(lib.dvf_photo_partials_floats(B, H, W, V)), device=tgt.device)
        # algorithmic bytes per target pixel: depth + C target + C*V sources (+ V masks)
        fwd_bytes = float(B * H * W) * (4 + 4 * C + 4 * C * V + (4 * V if mask is not None else 0))
        with L.timed("photo_fwd", 0.0, fwd_bytes):
            L.check(lib.dvf_photo_loss_fwd(L.dev(tgt, "tgt"), L.ptr_array(srcs, "srcs"), V, L.dev(depth, "depth"),
                                           L.dev(pose, "pose"), L.dev(K, "intrinsics"), L.dev(Kinv, "intrinsics_inv"),
                                           L.dev(mask, "mask"), L.dev(out), L.dev(out[1:]), L.dev(partials),
                                           B, C, H, W, in_scale, flags, L.stream()), "dvf_photo_loss_fwd")
        ctx.fwd_bytes = fwd_bytes
        ctx.save_for_backward(tgt, depth, pose, K, Kinv, mask, *srcs)
        ctx.flags, ctx.in_scale = flags, in_scale
        ctx.view_loss = out[1:]
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        tgt, depth, pose, K, Kinv, mask, *srcs = ctx.saved_tensors
        B, C, H, W = tgt.shape
        V = len(srcs)
        need = ctx.needs_input_grad
        lib = L.lib()
        g_tgt = torch.empty_like(tgt) if need[0] else None
        g_depth = torch.empty_like(depth) if need[1] else None
        g_pose = torch.empty_like(pose) if need[2] else None
        g_mask = torch.empty_like(mask) if (mask is not None and need[5]) else None
        g_srcs = [torch.zeros_like(s) if need[7 + i] else None for i, s in enumerate(srcs)]
        ws = torch.empty(int(lib.dvf_photo_pose_ws_floats(B, H, W, V)), device=tgt.device) if need[2] else None
        gl = _f32c(gloss).reshape(1)
        # bwd = fwd (recompute) + grad depth + mask grads + grad target + RMW scatter into the source grads
        bwd_bytes = ctx.fwd_bytes + float(B * H * W) * (
            (4 if g_depth is not None else 0) + (4 * V if g_mask is not None else 0) +
            (4 * C if g_tgt is not None else 0) + 8 * C * sum(1 for g in g_srcs if g is not None))
        with L.timed("photo_bwd", 0.0, bwd_bytes):
            L.check(lib.dvf_photo_loss_bwd(L.dev(tgt), L.ptr_array(srcs), V, L.dev(depth), L.dev(pose), L.dev(K),
                                           L.dev(Kinv), L.dev(mask), L.dev(gl, "grad_loss"), L.dev(g_depth),
                                           L.dev(g_pose), L.dev(g_tgt), L.ptr_array(g_srcs), L.dev(g_mask), L.dev(ws),
                                           B, C, H, W, ctx.in_scale, ctx.flags, L.stream()), "dvf_photo_loss_bwd")
        return (g_tgt, g_depth, g_pose, None, None, g_mask, None, *g_srcs)


class SmoothLossFn(torch.autograd.Function):
    """smooth_loss over a list of maps (reference loss_functions.py:23-41, loss_functions_sfm.py:59-77):
    sum_s w_s * (mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2|), w_{s+1} = w_s / scale_factor."""

    @staticmethod
    def forward(ctx, scale_factor, *maps):
        maps = [_f32c(m) for m in maps]
        lib = L.lib()
        out = torch.empty(1, device=maps[0].device)
        weights, w = [], 1.0
        for i, m in enumerate(maps):
            B, C, H, W = m.shape
            partials = torch.empty(int(lib.dvf_smooth_partials_floats(B * C, H, W)), device=m.device)
            L.check(lib.dvf_smooth_loss_fwd(L.dev(m, "pred_map"), L.dev(out), L.dev(partials), B * C, H, W, w,
                                            1 if i > 0 else 0, L.stream()), "dvf_smooth_loss_fwd")
            weights.append(w)
            w /= scale_factor
        ctx.save_for_backward(*maps)
        ctx.weights = weights
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        maps = ctx.saved_tensors
        gl = _f32c(gloss).reshape(1)
        grads = []
        for i, (m, w) in enumerate(zip(maps, ctx.weights)):
            if not ctx.needs_input_grad[1 + i]:
                grads.append(None)
                continue
            B, C, H, W = m.shape
            g = torch.empty_like(m)
            L.check(L.lib().dvf_smooth_loss_bwd(L.dev(m), L.dev(gl, "grad_loss"), L.dev(g), B * C, H, W, w,
                                                L.stream()), "dvf_smooth_loss_bwd")
            grads.append(g)
        return (None, *grads)


class EdgeSmoothLossFn(torch.autograd.Function):
    """Edge-aware first-order smoothness of the inverse depth as the reference's Caffe graph defines it
    (experiments/depth_odometry_feature/train.prototxt:4452-4661): see csrc/paper_losses.hip."""

    @staticmethod
    def forward(ctx, inv_depth, img, in_scale, edge_k, weight):
        inv_depth, img = _f32c(inv_depth), _f32c(img)
        B, C, H, W = img.shape
        if tuple(inv_depth.shape) not in ((B, 1, H, W), (B, H, W)):
            raise ValueError(f"inv_depth {tuple(inv_depth.shape)} does not match image {tuple(img.shape)}")
        lib = L.lib()
        out = torch.empty(1, device=img.device)
        partials = torch.empty(int(lib.dvf_edge_smooth_partials_floats(B, H, W)), device=img.device)
        L.check(lib.dvf_edge_smooth_fwd(L.dev(inv_depth, "inv_depth"), L.dev(img, "img"), L.dev(out), L.dev(partials), B, C, H, W,
                                        in_scale, edge_k, weight, 0, L.stream()), "dvf_edge_smooth_fwd")
        ctx.save_for_backward(inv_depth, img)
        ctx.cfg = (in_scale, edge_k, weight)
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        inv_depth, img = ctx.saved_tensors
        B, C, H, W = img.shape
        g = torch.empty_like(inv_depth)
        in_scale, edge_k, weight = ctx.cfg
        L.check(L.lib().dvf_edge_smooth_bwd(L.dev(inv_depth), L.dev(img), L.dev(_f32c(gloss).reshape(1), "grad_loss"), L.dev(g),
                                            B, C, H, W, in_scale, edge_k, weight, L.stream()), "dvf_edge_smooth_bwd")
        return g, None, None, None, None


class ExplainabilityLossFn(torch.autograd.Function):
    """explainability_loss over a list of masks (reference loss_functions_sfm.py:49-56)."""

    @staticmethod
    def forward(ctx, *masks):
        masks = [_f32c(m) for m in masks]
        out = torch.empty(1, device=masks[0].device)
        partials = torch.empty(1024, device=masks[0].device)
        for i, m in enumerate(masks):
            L.check(L.lib().dvf_bce_ones_fwd(L.dev(m, "mask"), L.dev(out), L.dev(partials), m.numel(), 1 if i > 0 else 0,
                                             L.stream()), "dvf_bce_ones_fwd")
        ctx.save_for_backward(*masks)
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        gl = _f32c(gloss).reshape(1)
        grads = []
        for i, m in enumerate(ctx.saved_tensors):
            if not ctx.needs_input_grad[i]:
                grads.append(None)
                continue
            g = torch.empty_like(m)
            L.check(L.lib().dvf_bce_ones_bwd(L.dev(m), L.dev(gl, "grad_loss"), L.dev(g), m.numel(), L.stream()),
                    "dvf_bce_ones_bwd")
            grads.append(g)
        return tuple(grads)


class PoseVec2MatFn(torch.autograd.Function):
    """pose_vec2mat (reference inverse_warp.py:141-157): [n,6] -> [n,3,4]."""

    @staticmethod
    def forward(ctx, vec, flags):
        vec = _f32c(vec)
        n = vec.shape[0]
        out = torch.empty((n, 3, 4), device=vec.device, dtype=torch.float32)
        L.check(L.lib().dvf_pose_vec2mat_fwd(L.dev(vec, "vec"), L.dev(out), n, flags, L.stream()), "dvf_pose_vec2mat_fwd")
        ctx.save_for_backward(vec)
        ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, gmat):
        (vec,) = ctx.saved_tensors
        n = vec.shape[0]
        g = torch.empty_like(vec)
        ws = torch.empty(n * 12, device=vec.device)
        L.check(L.lib().dvf_pose_vec2mat_bwd(L.dev(vec), L.dev(_f32c(gmat), "grad_out"), L.dev(g), L.dev(ws), n, ctx.flags,
                                             L.stream()), "dvf_pose_vec2mat_bwd")
        return g, None


class Pixel2CamFn(torch.autograd.Function):
    """pixel2cam (reference inverse_warp.py:26-40)."""

    @staticmethod
    def forward(ctx, depth, Kinv):
        depth, Kinv = _f32c(depth), _f32c(Kinv)
        B, H, W = depth.shape
        cam = torch.empty((B, 3, H, W), device=depth.device, dtype=torch.float32)
        L.check(L.lib().dvf_pixel2cam_fwd(L.dev(depth, "depth"), L.dev(Kinv, "intrinsics_inv"), L.dev(cam), B, H, W,
                                          L.stream()), "dvf_pixel2cam_fwd")
        ctx.save_for_backward(Kinv)
        ctx.shape = (B, H, W)
        return cam

    @staticmethod
    def backward(ctx, gcam):
        (Kinv,) = ctx.saved_tensors
        B, H, W = ctx.shape
        g = torch.empty((B, H, W), device=gcam.device, dtype=torch.float32)
        L.check(L.lib().dvf_pixel2cam_bwd(L.dev(Kinv), L.dev(_f32c(gcam), "grad_out"), L.dev(g), B, H, W, L.stream()),
                "dvf_pixel2cam_bwd")
        return g, None


class Cam2PixelFn(torch.autograd.Function):
    """cam2pixel (reference inverse_warp.py:43-74): cam [B,3,H,W] -> normalised grid [B,H,W,2]."""

    @staticmethod
    def forward(ctx, cam, rot, tr, flags):
        cam = _f32c(cam)
        rot = _f32c(rot) if rot is not None else None
        tr = _f32c(tr).reshape(-1, 3) if tr is not None else None
        B, _, H, W = cam.shape
        grid = torch.empty((B, H, W, 2), device=cam.device, dtype=torch.float32)
        L.check(L.lib().dvf_cam2pixel_fwd(L.dev(cam, "cam_coords"), L.dev(rot, "proj_c2p_rot"), L.dev(tr, "proj_c2p_tr"),
                                          L.dev(grid), B, H, W, flags, L.stream()), "dvf_cam2pixel_fwd")
        ctx.save_for_backward(cam, rot, tr)
        ctx.flags = flags
        return grid

    @staticmethod
    def backward(ctx, ggrid):
        cam, rot, tr = ctx.saved_tensors
        B, _, H, W = cam.shape
        gcam = torch.empty_like(cam) if ctx.needs_input_grad[0] else None
        need_rt = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        ws = torch.empty((B, 12), device=cam.device) if need_rt else None
        L.check(L.lib().dvf_cam2pixel_bwd(L.dev(cam), L.dev(rot), L.dev(tr), L.dev(_f32c(ggrid), "grad_out"), L.dev(gcam),
                                          L.dev(ws), B, H, W, ctx.flags, L.stream()), "dvf_cam2pixel_bwd")
        g_rot = ws[:, 3:].reshape(B, 3, 3) if (rot is not None and ctx.needs_input_grad[1]) else None
        g_tr = ws[:, :3].reshape(B, 3, 1) if (tr is not None and ctx.needs_input_grad[2]) else None
        return gcam, g_rot, g_tr, None
