/*
 * dvf_hip.h -- C ABI of libdvf_hip.so: the MI355X (gfx950) kernels behind the
 * depth + visual-odometry training hot path of Depth-VO-Feat (pytorch_version/).
 *
 * Conventions (every entry point):
 *   - plain C, no torch types; every pointer is a DEVICE pointer to fp32 data unless
 *     the parameter is documented as a host array;
 *   - tensors are contiguous NCHW (the reference's layout);
 *   - `stream` is a hipStream_t passed as void*; the call only enqueues work on it
 *     (no allocation, no synchronisation, capturable in a hipGraph);
 *   - workspaces are passed in by the caller; sizes come from the *_workspace_floats
 *     query functions;
 *   - the return value is 0 (DVF_OK) or a negative DVF_ERR_* code; nothing throws.
 *
 * The host side that binds these (ctypes, `depth-vo-feat_amd/dvf/lib.py`) mirrors the
 * reference's Python API one to one; the reference call site each entry replaces is
 * cited per function as pytorch_version/<file>:<line>.
 */
#ifndef DVF_HIP_H
#define DVF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVF_OK 0
#define DVF_ERR_INVALID_ARG (-1)
#define DVF_ERR_LAUNCH (-2)
#define DVF_ERR_UNSUPPORTED (-3)

/* geometry flags (bit field) */
#define DVF_ROT_QUAT 1u      /* pose[3:6] are the last 3 quaternion coefficients (inverse_warp.py:117-138); default euler (:77-114) */
#define DVF_PAD_BORDER 2u    /* padding_mode='border'; default 'zeros' incl. the coords-set-to-2 overwrite (inverse_warp.py:67-71) */
#define DVF_ALIGN_CORNERS 4u /* grid_sample(align_corners=True); default False = what the reference runs as (SURVEY.md preamble #5) */

/* activation codes for the convolution epilogues */
#define DVF_ACT_NONE 0
#define DVF_ACT_RELU 1
#define DVF_ACT_SIGMOID_AFFINE 2 /* alpha * sigmoid(x) + beta  (DispNetS.py:112; masks use alpha=1, beta=0) */

#define DVF_MAX_VIEWS 4
#define DVF_MAX_SEGS 3

int dvf_version(void);
const char *dvf_error_string(int code);

/* ---------------------------------------------------------------- inverse warp (image out)
 * Replaces inverse_warp.inverse_warp (pytorch_version/inverse_warp.py:160-193; copy at
 * loss_functions.py:198-231): pixel2cam -> pose_vec2mat -> K@[R|t] -> cam2pixel -> grid_sample.
 *   img [B,C,H,W], depth [B,H,W], pose [B,6] (tx,ty,tz,rx,ry,rz), K/Kinv [B,3,3] -> out [B,C,H,W]. */
int dvf_inverse_warp_fwd(const float *img, const float *depth, const float *pose, const float *K,
                         const float *Kinv, float *out, int B, int C, int H, int W, uint32_t flags,
                         void *stream);
/* Backward of the above for an incoming grad_out [B,C,H,W].  Any of g_img / g_depth / g_pose may be
 * NULL (not needed).  g_img must be ZEROED by the caller (scatter-add target).  pose_ws: >=
 * dvf_pose_ws_floats(1, B) floats, zeroed by this call. */
int dvf_inverse_warp_bwd(const float *img, const float *depth, const float *pose, const float *K,
                         const float *Kinv, const float *grad_out, float *g_img, float *g_depth,
                         float *g_pose, float *pose_ws, int B, int C, int H, int W, uint32_t flags,
                         void *stream);
int64_t dvf_pose_ws_floats(int V, int B);

/* ---------------------------------------------------------------- fused warp + photometric L1
 * Replaces the body of loss_functions.photometric_reconstruction_loss (loss_functions.py:7-20) and of
 * one_scale() in loss_functions_sfm.photometric_reconstruction_loss (loss_functions_sfm.py:10-36):
 * for every view v: warp src_v with pose_v, exact-zero mask, optional explainability mask, |.|, mean
 * over B*C*H*W; the views are summed.  One kernel per pyramid scale handles all V views of a pixel.
 *   tgt [B,C,H,W]; srcs: HOST array of V device pointers, each [B,C,H,W]; depth [B,H,W];
 *   pose [V,B,6]; K/Kinv [B,3,3]; mask NULL or [B,V,H,W];
 *   loss_out: 1 float (sum over views), view_loss: NULL or V floats;
 *   partials: workspace >= dvf_photo_partials_floats(B,H,W,V) floats. */
int dvf_photo_loss_fwd(const float *tgt, const float *const *srcs, int V, const float *depth,
                       const float *pose, const float *K, const float *Kinv, const float *mask,
                       float *loss_out, float *view_loss, float *partials, int B, int C, int H, int W,
                       uint32_t flags, void *stream);
int64_t dvf_photo_partials_floats(int B, int H, int W, int V);
/* Backward: grad_loss is a DEVICE scalar (upstream gradient).  Outputs (each may be NULL):
 *   g_depth [B,H,W] (written), g_pose [V,B,6] (written), g_tgt [B,C,H,W] (written),
 *   g_srcs: HOST array of V device pointers or NULL entries, each [B,C,H,W], ZEROED by the caller,
 *   g_mask [B,V,H,W] (written).  pose_ws >= dvf_pose_ws_floats(V,B). */
int dvf_photo_loss_bwd(const float *tgt, const float *const *srcs, int V, const float *depth,
                       const float *pose, const float *K, const float *Kinv, const float *mask,
                       const float *grad_loss, float *g_depth, float *g_pose, float *g_tgt,
                       float *const *g_srcs, float *g_mask, float *pose_ws, int B, int C, int H, int W,
                       uint32_t flags, void *stream);

/* ---------------------------------------------------------------- smoothness loss
 * Replaces one map of smooth_loss (loss_functions.py:23-41 ; loss_functions_sfm.py:59-77):
 * weight * (mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2|) of a [N,H,W] stack of planes (N = B*C).
 * loss_out[0] (+)= result: accumulate != 0 adds to the value already there (multi-scale lists). */
int dvf_smooth_loss_fwd(const float *map, float *loss_out, float *partials, int N, int H, int W,
                        float weight, int accumulate, void *stream);
int64_t dvf_smooth_partials_floats(int N, int H, int W);
int dvf_smooth_loss_bwd(const float *map, const float *grad_loss, float *g_map, int N, int H, int W,
                        float weight, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVF_HIP_H */
