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
#define DVF_POSE_SE3 8u      /* pose = (wx,wy,wz,ux,uy,uz): exponential map with t = R u (se3_generate.py:7-53, caffe/python/pygeometry.py:6-114);
                                the front end of unsupervise_dvo.py:98-100 */
#define DVF_PIXEL_COORDS 16u /* sample at pixel coordinates u = fx X/(Z+1e-12) + cx, taps bounds-checked individually, no coords
                                overwrite and no depth clamp: Caffe GeoTransform -> PinHole -> InverseWarping semantics
                                (geometry_transformation.cu:10-47, pin_hole_layer.cu:10-50, inverse_warping_layer.cu:10-52) */

#define DVF_CAFFE_ABSLOSS 32u /* photometric loss normalised and differentiated like Caffe's AbsLoss (abs_loss_layer.cu:10-34,
                                experiments/depth_odometry_feature/train.prototxt:4428-4446): sum |warped - target| / B (per-sample
                                sum, not a mean), NO exact-zero validity mask, d|d|/dd = (d > 0) - (d <= 0) */

/* activation codes for the convolution epilogues */
#define DVF_ACT_NONE 0
#define DVF_ACT_RELU 1
#define DVF_ACT_SIGMOID_AFFINE 2 /* alpha * sigmoid(x) + beta  (DispNetS.py:112; masks use alpha=1, beta=0) */

#define DVF_MAX_VIEWS 4
#define DVF_MAX_SEGS 5

int dvf_version(void);
const char *dvf_error_string(int code);
/* 1 when the library was built with -DDVF_TUNING (tuning knobs read from the environment, kernel ablation switches);
 * the product build returns 0: it reads no environment variable and cannot be switched into computing partial results. */
int dvf_build_has_tuning(void);
/* Which kernels the LAST dvf_conv2d_* call of this thread launched: records of 12 ints
 *   {kernel, MT, NT, WM, CK, TBU, KS, BN, NST, threads, lds_bytes, mode}   (kernel: 1 conv_pipe, 2 conv_gather,
 *   3 head_fwd {C_out, nseg}, 4 head_dgrad, 5 head_wgrad, 6 conv_wgrad {MT, NTW, prefetch, CK, BH, PSPLIT, vec_p, stride,
 *   threads, lds, KH*KW}, 7 head_seg_dgrad) copied to the HOST array `out`; returns the number of ints written.
 * The planner is shape dependent; tests use this to show that every plan of the benchmark step is parity-tested. */
int dvf_conv2d_last_plans(int *out, int max_ints);

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

/* ---------------------------------------------------------------- stand-alone geometry helpers
 * The public building blocks of inverse_warp.py, for callers that use them directly (the fused kernels above do
 * not go through these).
 *   pose_vec2mat (inverse_warp.py:141-157; euler2mat :77-114, quat2mat :117-138): pose [n,6] -> [n,3,4];
 *     backward: g_mat [n,3,4] -> g_pose [n,6], ws = n*12 floats.
 *   pixel2cam (:26-40): depth [B,H,W], Kinv [B,3,3] -> cam [B,3,H,W]; backward -> g_depth.
 *   cam2pixel (:43-74): cam [B,3,H,W], rot [B,3,3] or NULL, tr [B,3] or NULL -> grid [B,H,W,2] (x_n, y_n);
 *     backward: g_cam [B,3,H,W] (may be NULL) and g_rot_tr_ws [B,12] = (g_tr[3], g_rot[9]) (may be NULL). */
int dvf_pose_vec2mat_fwd(const float *pose, float *out, int n, uint32_t flags, void *stream);
int dvf_pose_vec2mat_bwd(const float *pose, const float *g_mat, float *g_pose, float *ws, int n, uint32_t flags,
                         void *stream);
int dvf_pixel2cam_fwd(const float *depth, const float *Kinv, float *cam, int B, int H, int W, void *stream);
int dvf_pixel2cam_bwd(const float *Kinv, const float *g_cam, float *g_depth, int B, int H, int W, void *stream);
int dvf_cam2pixel_fwd(const float *cam, const float *rot, const float *tr, float *grid, int B, int H, int W,
                      uint32_t flags, void *stream);
int dvf_cam2pixel_bwd(const float *cam, const float *rot, const float *tr, const float *g_grid, float *g_cam,
                      float *g_rot_tr_ws, int B, int H, int W, uint32_t flags, void *stream);

/* ---------------------------------------------------------------- fused warp + photometric L1
 * Replaces the body of loss_functions.photometric_reconstruction_loss (loss_functions.py:7-20) and of
 * one_scale() in loss_functions_sfm.photometric_reconstruction_loss (loss_functions_sfm.py:10-36):
 * for every view v: warp src_v with pose_v, exact-zero mask, optional explainability mask, |.|, mean
 * over B*C*H*W; the views are summed.  One kernel per pyramid scale handles all V views of a pixel; the 2x2
 * source neighbourhoods of a 64x4 target tile are staged through an LDS footprint tile.
 *   tgt [B,C,H,W]; srcs: HOST array of V device pointers, each [B,C,H,W]; depth [B,H,W];
 *   pose [V,B,6]; K/Kinv [B,3,3]; mask NULL or [B,V,H,W];
 *   in_scale: every tgt / src value is used as fl(in_scale * x) -- the `0.004 * img` of unsupervise.py:101 without
 *     materialising scaled copies (1.0f = the reference signature; must not be 0);
 *   loss_out: 1 float (sum over views), view_loss: NULL or V floats;
 *   partials: workspace >= dvf_photo_partials_floats(B,H,W,V) floats. */
int dvf_photo_loss_fwd(const float *tgt, const float *const *srcs, int V, const float *depth,
                       const float *pose, const float *K, const float *Kinv, const float *mask,
                       float *loss_out, float *view_loss, float *partials, int B, int C, int H, int W,
                       float in_scale, uint32_t flags, void *stream);
int64_t dvf_photo_partials_floats(int B, int H, int W, int V);
/* Backward: grad_loss is a DEVICE scalar (upstream gradient).  Outputs (each may be NULL):
 *   g_depth [B,H,W] (written), g_pose [V,B,6] (written), g_tgt [B,C,H,W] (written),
 *   g_srcs: HOST array of V device pointers or NULL entries, each [B,C,H,W], ZEROED by the caller
 *   (scatter-add target: accumulated per block in LDS, flushed with row-contiguous atomics),
 *   g_mask [B,V,H,W] (written).  pose_ws >= dvf_photo_pose_ws_floats(B,H,W,V) floats (per-block [R|t] partials,
 *   summed in fixed order: the pose gradient is bit-reproducible).  g_tgt / g_srcs need C <= 32
 *   (DVF_ERR_UNSUPPORTED otherwise; the reference's feature maps have 32 channels, feat_extractor.py:13-36). */
int dvf_photo_loss_bwd(const float *tgt, const float *const *srcs, int V, const float *depth,
                       const float *pose, const float *K, const float *Kinv, const float *mask,
                       const float *grad_loss, float *g_depth, float *g_pose, float *g_tgt,
                       float *const *g_srcs, float *g_mask, float *pose_ws, int B, int C, int H, int W,
                       float in_scale, uint32_t flags, void *stream);
int64_t dvf_photo_pose_ws_floats(int B, int H, int W, int V);

/* ---------------------------------------------------------------- input pipeline: frame -> network input on the GPU
 * Replaces, per frame, `imresize(imread(f).astype(np.float32), (H, W))` of the reference's loaders (un_dataset.py:63-66,
 * dataset.py:50-51; scipy.misc.imresize = bytescale [min,max]->[0,255] + PIL BILINEAR on uint8), bit for bit:
 *   src_hwc uint8 [IH,IW,C] (the decoded frame) -> dst_chw float32 [C,OH,OW] (integer values 0..255).
 * hbounds/hcoef [OW,2]/[OW,hksize] and vbounds/vcoef [OH,2]/[OH,vksize]: DEVICE int32 tables of PIL's separable triangle
 * filter in 22-bit fixed point (dvf/image_ops.py builds them as Pillow's precompute_coeffs / normalize_coeffs_8bpc do);
 * tmp: uint8 [IH,OW,C] workspace; minmax: 2 int32 workspace. */
int dvf_imresize_u8(const uint8_t *src_hwc, int IH, int IW, int C, const int *hbounds, const int *hcoef, int hksize,
                    const int *vbounds, const int *vcoef, int vksize, uint8_t *tmp, int *minmax, float *dst_chw, int OH, int OW,
                    void *stream);

/* ---------------------------------------------------------------- edge-aware smoothness (paper / Caffe graph variant)
 * experiments/depth_odometry_feature/train.prototxt:4452-4661 with caffe/include/caffe/filler.hpp:267-316 and
 * abs_loss_layer.cu:10-34: loss_out[0] (+)= weight * (sum |exp(-k sum_c |d_y I_c|) d_y D| + sum |exp(-k sum_c |d_x I_c|) d_x D|) / B
 * over the (H-2) x (W-2) interior (3x3 valid central differences, +-0.5), I = in_scale * img [B,C,H,W], D = inv_depth [B,H,W];
 * backward writes g_inv_depth [B,H,W] (AbsLoss sign convention: zero counts as positive).  Parity unpinned (no Caffe here). */
int dvf_edge_smooth_fwd(const float *inv_depth, const float *img, float *loss_out, float *partials, int B, int C, int H, int W,
                        float in_scale, float edge_k, float weight, int accumulate, void *stream);
int64_t dvf_edge_smooth_partials_floats(int B, int H, int W);
int dvf_edge_smooth_bwd(const float *inv_depth, const float *img, const float *grad_loss, float *g_inv_depth, int B, int C, int H,
                        int W, float in_scale, float edge_k, float weight, void *stream);

/* ---------------------------------------------------------------- smoothness loss
 * Replaces one map of smooth_loss (loss_functions.py:23-41 ; loss_functions_sfm.py:59-77):
 * weight * (mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2|) of a [N,H,W] stack of planes (N = B*C).
 * loss_out[0] (+)= result: accumulate != 0 adds to the value already there (multi-scale lists). */
int dvf_smooth_loss_fwd(const float *map, float *loss_out, float *partials, int N, int H, int W,
                        float weight, int accumulate, void *stream);
int64_t dvf_smooth_partials_floats(int N, int H, int W);
int dvf_smooth_loss_bwd(const float *map, const float *grad_loss, float *g_map, int N, int H, int W,
                        float weight, void *stream);

/* ---------------------------------------------------------------- convolutions (fp32 MFMA)
 * One descriptor for nn.Conv2d and nn.ConvTranspose2d as the reference's networks use them
 * (DispNetS.py:7-34, PoseExpNet_sfm.py:6-17, feat_extractor.py:6-41): square-ish kernels up to 7x7,
 * stride 1 or 2, symmetric padding, groups = 1.  H_out/W_out are the sizes actually kept, i.e. AFTER
 * crop_like (DispNetS.py:37-39): the kernels never compute cropped-away pixels.
 * Weight layouts are torch's: Conv2d [C_out,C_in,KH,KW]; ConvTranspose2d [C_in,C_out,KH,KW].
 * The input may be a VIRTUAL CONCAT of up to DVF_MAX_SEGS tensors [N,seg_channels[i],H_in,W_in]
 * (replaces torch.cat at DispNetS.py:98-128): in_segs / din_segs are HOST arrays of device pointers. */
typedef struct dvf_conv_desc {
    int N, C_in, H_in, W_in, C_out, H_out, W_out, KH, KW, stride, pad, transposed;
    int act;           /* DVF_ACT_*: fused epilogue of the forward */
    float alpha, beta; /* DVF_ACT_SIGMOID_AFFINE parameters */
} dvf_conv_desc;

/* out = act(conv(cat(in_segs), w) + bias).  Replaces nn.Conv2d / nn.ConvTranspose2d (+ReLU / Sigmoid, and the
 * alpha*sigmoid+beta of DispNetS.py:112) forward.  bias may be NULL. */
int dvf_conv2d_fwd(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                   const float *w, const float *bias, float *out, void *stream);
/* d(loss)/d(input segment i) for i with din_segs[i] != NULL, given dpre = d(loss)/d(pre-activation output). */
int dvf_conv2d_dgrad(const dvf_conv_desc *d, const float *dpre, const float *w, float *const *din_segs,
                     const int *seg_channels, int nseg, void *stream);
/* d(loss)/d(w) in w's own layout; accumulate != 0 adds to dw, otherwise dw is zeroed first. */
int dvf_conv2d_wgrad(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                     const float *dpre, float *dw, int accumulate, void *stream);
/* dvf_conv2d_wgrad + the bias gradient of the same layer, dbias[c] (+)= sum over (n, y, x) of dpre[n][c][y][x]
 * (torch: the bias output of ConvolutionBackward).  For Conv2d layers the pipelined weight-gradient kernel multiplies its
 * dpre tile by one more column -- of ones -- so no extra pass reads dpre; otherwise one reduction pass runs after it. */
int dvf_conv2d_wgrad_bias(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                          const float *dpre, float *dw, int accumulate, float *dbias, int accumulate_dbias, void *stream);
/* Run-to-run DETERMINISTIC weight (+ bias, dbias may be NULL) gradient -- the reference's CPU path is (torch CPU
 * ConvolutionBackward, train.py:213).  The pipelined kernel splits the pixel reduction over all CUs; with a scratch buffer of
 * dvf_conv2d_wgrad_ws_floats(...) floats every block stores its partial tiles and one pass adds them in block order, instead
 * of float atomics in order of arrival (ws == NULL or too small: the atomic form of dvf_conv2d_wgrad / _wgrad_bias). */
int64_t dvf_conv2d_wgrad_ws_floats(const dvf_conv_desc *d, const int *seg_channels, int nseg);
int dvf_conv2d_wgrad_det(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                         const float *dpre, float *dw, int accumulate, float *dbias, int accumulate_dbias, float *ws,
                         int64_t ws_floats, void *stream);

/* Packed-weight fast path: LDS-DMA pipelined kernels that read the weights from a pre-packed copy (the exact LDS
 * image of every reduction chunk), same results as dvf_conv2d_fwd / dvf_conv2d_dgrad.  op_kind 0 = forward,
 * 1 = dgrad.  Protocol: n = dvf_conv2d_packed_floats(...) (negative = DVF_ERR_*; DVF_ERR_UNSUPPORTED means "use the
 * unpacked entry for this geometry"); the caller allocates n floats and ZERO-FILLS them once; dvf_conv2d_pack()
 * refreshes the copy whenever w changed (after every optimizer step, train.py:214); the *_packed entries consume it.
 * The packing depends on the whole descriptor (incl. N, H, W) and on seg_channels: use the same ones everywhere. */
int64_t dvf_conv2d_packed_floats(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind);
int dvf_conv2d_pack(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind, const float *w, float *packed,
                    void *stream);
int dvf_conv2d_fwd_packed(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                          const float *packed, const float *bias, float *out, float *ws, int64_t ws_floats, void *stream);
/* (dgrad: input segments the pipelined kernel does not take -- at most 32 channels -- have no share in the packed copy
 * and are computed from the unpacked weights w, which may be NULL only if every segment is packed) */
int dvf_conv2d_dgrad_packed(const dvf_conv_desc *d, const float *dpre, const float *packed, const float *w,
                            float *const *din_segs, const int *seg_channels, int nseg, float *ws, int64_t ws_floats,
                            void *stream);
/* dgrad that also finishes the PRODUCING layers' backward pass.  mask_segs[s] != NULL says input segment s is the output
 * y of a ReLU convolution (reference: nn.ReLU(inplace=True) after every conv, DispNetS.py:10-16,30-41, PoseExpNet_sfm.py:9-13):
 * din_segs[s] leaves as (y > 0) ? din : 0 -- which is dL/dpre of that layer when every consumer of y does the same
 * (the ReLU mask distributes over the sum autograd forms) -- and its sums over (n, y, x) are ADDED to dbias_segs[s][c]
 * (that layer's bias gradient; float atomics; may be NULL).  packed may be NULL (unpacked kernels only); mask_segs entries
 * may be NULL (plain dgrad of that segment).  Replaces torch's ThresholdBackward + the bias reduction of ConvolutionBackward. */
int dvf_conv2d_dgrad_masked(const dvf_conv_desc *d, const float *dpre, const float *packed, const float *w,
                            float *const *din_segs, const int *seg_channels, int nseg, float *ws, int64_t ws_floats,
                            const float *const *mask_segs, float *const *dbias_segs, void *stream);
/* The unpacked entries with a split-K workspace (>= dvf_conv2d_ws_floats(...) floats; NULL = the float-atomic fallback of
 * dvf_conv2d_fwd / dvf_conv2d_dgrad): small grids split the reduction over blocks, and with the workspace the partial
 * tiles are summed in a fixed order -- activations are then bit-reproducible from run to run. */
int dvf_conv2d_fwd_ws(const dvf_conv_desc *d, const float *const *in_segs, const int *seg_channels, int nseg,
                      const float *w, const float *bias, float *out, float *ws, int64_t ws_floats, void *stream);
int dvf_conv2d_dgrad_ws(const dvf_conv_desc *d, const float *dpre, const float *w, float *const *din_segs,
                        const int *seg_channels, int nseg, float *ws, int64_t ws_floats, void *stream);
/* Optional split-K workspace of the packed entries: with ws (>= dvf_conv2d_ws_floats(...) floats, contents
 * irrelevant) small grids split the reduction over blocks that store plain partial tiles which one pass reduces
 * (+bias, activation); with ws == NULL they accumulate with float atomics into a zeroed output instead. */
int64_t dvf_conv2d_ws_floats(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind);
/* Batched packing (one launch per optimizer step instead of one per convolution): dvf_conv2d_pack_jobs() writes the
 * job records of one convolution (1 for op_kind 0, nseg for op_kind 1; DVF_PACK_JOB_BYTES each, opaque) into HOST
 * memory and their block counts into blocks_out, returning the number of jobs (or a negative DVF_ERR_*).  The caller
 * concatenates the records of all convolutions, uploads them and the exclusive prefix sum of the block counts
 * (njobs + 1 ints) to the device once, and calls dvf_conv2d_pack_batch() after every optimizer step with lds_bytes = the
 * maximum that the pack_jobs calls left in *lds_bytes_out (they only ever raise it).  block_job_dev (optional, may be
 * NULL): total_blocks ints, the job index of every block -- saves each block a binary search through the prefix. */
#define DVF_PACK_JOB_BYTES 512
int dvf_conv2d_pack_jobs(const dvf_conv_desc *d, const int *seg_channels, int nseg, int op_kind, const float *w, float *packed,
                         void *jobs_host, int max_jobs, int *blocks_out, int *lds_bytes_out);
int dvf_conv2d_pack_batch(const void *jobs_dev, const int *block_prefix_dev, const int *block_job_dev, int njobs,
                          int total_blocks, int lds_bytes, void *stream);
/* Backward of the fused activation and of the bias in one pass: dpre = dy * act'(y) (y = the forward's
 * output, [N,C,HW]); dbias[c] = sum dpre (zeroed by the call).  dpre or dbias may be NULL. */
int dvf_act_bwd(const float *dy, const float *y, float *dpre, float *dbias, int N, int C, int HW, int act,
                float alpha, float beta, void *stream);
/* Same, with accumulate_dbias != 0: dbias is NOT zeroed first, the channel sums are added to it (a bias gradient that
 * lives in a zeroed gradient arena needs neither a memset nor a separate add). */
int dvf_act_bwd2(const float *dy, const float *y, float *dpre, float *dbias, int N, int C, int HW, int act, float alpha,
                 float beta, int accumulate_dbias, void *stream);
/* dvf_act_bwd2 with a run-to-run deterministic bias gradient: per-block partial sums go through ws
 * (>= dvf_act_bwd_ws_floats(N, C, HW) floats) and are added in index order instead of float atomics on dbias[c]. */
int64_t dvf_act_bwd_ws_floats(int N, int C, int HW);
int dvf_act_bwd_det(const float *dy, const float *y, float *dpre, float *dbias, int N, int C, int HW, int act, float alpha,
                    float beta, int accumulate_dbias, float *ws, int64_t ws_floats, void *stream);

/* ---------------------------------------------------------------- memory-bound helpers
 * planes = N*C throughout. */
/* F.interpolate(mode='bilinear', align_corners=False): out[Y][X] for Y<OH, X<OW with source scale
 * scale_h = H/(full OH) (0.5 for x2 up, DispNetS.py:115,121,127 -- OH/OW may be the crop_like-cropped size;
 * 2.0 for the x0.5 image pyramid of feat_extractor.py:44-46). */
int dvf_resize_bilinear_fwd(const float *in, float *out, int planes, int H, int W, int OH, int OW, float scale_h,
                            float scale_w, void *stream);
/* backward of the x2 case (gather form, no atomics): gout [planes,OH,OW] -> gin [planes,H,W] */
int dvf_upsample2x_bwd(const float *gout, float *gin, int planes, int H, int W, int OH, int OW, void *stream);
/* y = 1 / (x + eps)  (train.py:188 eps=0; unsupervise.py:99 eps=1e-4) and its backward gx = -gy * y^2 */
int dvf_recip_fwd(const float *x, float *y, float eps, int64_t n, void *stream);
int dvf_recip_bwd(const float *gy, const float *y, float *gx, int64_t n, void *stream);
/* out[plane] = scale * mean(in[plane])  (PoseExpNet_sfm.py:72-73: mean(3).mean(2) * 0.01) */
int dvf_spatial_mean_fwd(const float *in, float *out, int planes, int HW, float scale, void *stream);
int dvf_spatial_mean_bwd(const float *gout, float *gin, int planes, int HW, float scale, void *stream);
/* F.interpolate(mode='area') = adaptive average pooling (loss_functions_sfm.py:18-19) */
int dvf_area_downsample(const float *in, float *out, int planes, int H, int W, int OH, int OW, void *stream);
/* Depthwise nn.ConvTranspose2d(C, C, kernel_size=4, stride=2, padding=1, groups=C) of FeatExtractor's top-down path
 * (feat_extractor.py:38-41) with the residual add of :72-82 fused: out = skip + convT(x) + bias.  w [C,1,4,4];
 * x [N,C,H,W]; skip/out [N,C,2H,2W]; skip, bias may be NULL.  Backward: dx, dw (+db) from dy (d skip = dy). */
int dvf_dwconvt4x4s2_fwd(const float *x, const float *w, const float *bias, const float *skip, float *out, int N, int C,
                         int H, int W, void *stream);
int dvf_dwconvt4x4s2_bwd(const float *x, const float *w, const float *dy, float *dx, float *dw, float *db, int N, int C,
                         int H, int W, void *stream);
/* explainability_loss of one scale (loss_functions_sfm.py:49-56): loss_out[0] (+)= -mean(max(log mask, -100));
 * partials: workspace of >= 1024 floats.  Backward: g_mask = -grad_loss / (n * mask). */
int dvf_bce_ones_fwd(const float *mask, float *loss_out, float *partials, int64_t n, int accumulate, void *stream);
int dvf_bce_ones_bwd(const float *mask, const float *grad_loss, float *g_mask, int64_t n, void *stream);
/* torch.optim.Adam step (train.py:154-156, unsupervise.py:241) over a flat arena of n floats.
 * opt_state: DEVICE float[4] = {step, lr, step_size, bc2_sqrt}; the caller initialises {0, lr, 0, 0}.
 * advance_step != 0 increments step and refreshes the derived entries first (once per optimizer step).
 * grad_scale multiplies the gradient (1/world_size after a sum all-reduce). */
int dvf_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float *opt_state,
                  int advance_step, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVF_HIP_H */
