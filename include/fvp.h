/*
 * fvp.h - C ABI of the MI355X-native Faster-VoxelPose inference hot path.
 *
 * One flat `extern "C"` entry point per operator group of the reference's hot path
 * (heatmaps -> project_layer -> HDN -> JLN -> 3D joints; AlvinYH/Faster-VoxelPose,
 * lib/models).  The reference has no FFI layer of its own (it is pure PyTorch); each
 * declaration below names the reference Python site (file:line under the reference
 * checkout) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (PyTorch-ROCm storage);
 *     nothing is allocated, freed or retained by the library;
 *   - every call is asynchronous on the given `hipStream_t` (passed as void*; NULL = the
 *     null stream) and returns 0 or a hipError_t / FVP_E* code; no exceptions cross the ABI;
 *   - tensors are dense, row-major, fp32 unless stated; index tensors are int32 or int64
 *     as stated (int64 where the reference returns torch.int64);
 *   - thread-safe for calls on distinct streams with distinct output buffers.
 */
#ifndef FVP_H_
#define FVP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVP_ABI_VERSION 8
#define FVP_MAX_VIEWS 8
#define FVP_CAM_FLOATS 24 /* R[9] T[3] fx fy cx cy k[3] p[2] + 3 pad */
#define FVP_MAX_JOINTS 32

#define FVP_EINVAL 10001 /* bad argument (null pointer, unsupported size) */
#define FVP_ELIMIT 10002 /* size beyond a compiled limit (see message) */

typedef void* fvp_stream_t;

/* Per-dataset projection constants (lib/models/project_whole.py:49-60). */
typedef struct FvpGeom {
  float clamp_max; /* max(ORI_IMAGE_SIZE) : pixel clamp [-1, clamp_max]          :51 */
  float rt[6];     /* resize_transform 2x3, row-major                             :52 */
  float hm_w, hm_h;   /* HEATMAP_SIZE as floats                                   :53 */
  float img_w, img_h; /* IMAGE_SIZE as floats                                     :55 */
  int32_t W, H;       /* heatmap width / height                                       */
  int32_t V, J;       /* views, joints                                                */
  int32_t JP;         /* joints padded to a multiple of 4 (channels-last staging)     */
} FvpGeom;

int fvp_version(void);
/* 0 for the shipped library, which reads NO environment variable; 1 for the diagnostics build (-DFVP_DIAG=1,
 * tests/diag/libfvp_hip_diag.so - test / tool infrastructure) in which the kernel-selection, tuning and ablation
 * switches of DESIGN.md section 6 (FVP_* environment variables) are honoured. */
int fvp_diag_build(void);
/* sizeof(FvpGeom) (what = 0) / sizeof(FvpConvOp) (what = 1) as compiled into the library: lets a
 * binding check its struct mirrors before passing them. */
int fvp_sizeof(int what);
const char* fvp_error_string(int code);

/* ---- staging ------------------------------------------------------------------------
 * NCHW heatmaps [B,V,J,H,W] -> channels-last [B,V,H,W,JP] (pad channels = 0), so the four
 * bilinear taps of one (voxel, view) read JP contiguous floats each.  No reference
 * counterpart (F.grid_sample reads NCHW); this is the MI355X layout decision. */
int fvp_heatmaps_to_cl(const float* heat, float* heat_cl, int B, const FvpGeom* g, fvp_stream_t s);

/* ---- a-1/a-2/a-13: sampling grid ------------------------------------------------------
 * grid[v][ix*ny*nz + iy*nz + iz] = normalised (x,y) of voxel centre (ax[ix], ay[iy], az[iz])
 * in view v.  Replaces ProjectLayer.project_grid (project_whole.py:49-60,
 * project_individual.py:60-72) + cameras.project_point (utils/cameras.py:30-56) +
 * transforms.affine_transform_pts_cuda (utils/transforms.py:59-63).  The projection kernels
 * below evaluate the same device function on the fly; this export exists for the
 * drop-in `sample_grid` cache and for parity tests.  cams = [V][FVP_CAM_FLOATS]. */
int fvp_sample_grid(const float* ax, const float* ay, const float* az, int nx, int ny, int nz,
                    const float* cams, const FvpGeom* g, float* grid, fvp_stream_t s);

/* ---- a-3 (+ the z-max of a-4): whole-space back-projection ------------------------------
 * cubes[b][j][x][y][z] = clamp01(mean_v bilinear(heat_cl[b][v], grid(v, voxel)))  and/or
 * zmax[b][j][x][y] = max_z cubes.  Either output may be NULL.  Replaces
 * project_whole.ProjectLayer.forward (project_whole.py:62-88) and torch.max(x, dim=4)
 * (cnns_2d.py:174).  cams = [nsets][V][FVP_CAM_FLOATS]; frame_set[b] picks the camera set
 * (one per sequence, meta['seq'][b]). */
int fvp_project_whole(const float* heat_cl, const float* cams, const int32_t* frame_set,
                      const float* ax, const float* ay, const float* az, int X, int Y, int Z, int B,
                      const FvpGeom* g, float* cubes, float* zmax, fvp_stream_t s);

/* ---- a-8 without materialised cubes: the N proposal z-columns of every frame ------------------------
 * feat1d[b*N + k][j][z] = cubes[b][j][flat[b][k]][z] WITHOUT the cubes: the same device function as
 * fvp_project_whole evaluated on the N columns that human_detection_net.py:92-93 gathers (bit-equal to
 * the gather from materialised cubes; tested).  flat [B][N] int64 from fvp_nms_topk; an index outside
 * [0, X*Y) yields a zero column.  Used by the fused forward (HumanDetectionNet inside
 * FasterVoxelPoseNet.forward), where nothing else of the 4*J*X*Y*Z bytes per frame is ever read. */
int fvp_project_columns(const float* heat_cl, const float* cams, const int32_t* frame_set, const float* ax,
                        const float* ay, const float* az, int X, int Y, int Z, int B, const FvpGeom* g,
                        const int64_t* flat, int N, float* feat1d, fvp_stream_t s);

/* z-max of already materialised cubes [n][Z] -> [n] (n = B*J*X*Y): the first statement of
 * CenterNet.forward (cnns_2d.py:174) when it is called on its own. */
int fvp_zmax(const float* cubes, float* zmax, long n, int Z, fvp_stream_t s);

/* ---- a-14 integer part: per-person fine-grid windows ------------------------------------
 * For each of n proposals ([n][7] rows of proposal_centers): tl = round_half_even(c*scale+bias)
 * (int32), offset (mm), margin from the bbox, start/end clipped to the fine grid.  boxes =
 * [n][9] int32 = tl[3], start[3], end[3]; offset = [n][3].  Bit-exact with
 * project_individual.ProjectLayer.forward :110-121.  consts = scale[3], bias[3], whole[3],
 * ind[3] (12 floats, device memory: host-side values of project_individual.py:22-30 uploaded once);
 * fine_cube = fine[3], cube[3]: six int32 in HOST memory, read at call time and passed to the
 * kernel by value (the only host-memory pointer argument of this ABI besides FvpGeom / op lists). */
int fvp_person_boxes(const float* centers, int n, const float* consts, const int32_t* fine_cube,
                     int32_t* boxes, float* offset, fvp_stream_t s);

/* ---- a-14 sampling part (drop-in, materialises the cube) ---------------------------------
 * cubes[p][j][C][C][C]; person p lives in frame person_frame[p]; fine-grid axis tables
 * fx/fy/fz of lengths fine[0..2].  Windows outside [start,end) stay 0; persons with
 * person_valid[p]==0 (may be NULL = all valid) are zero-filled.  Replaces
 * project_individual.ProjectLayer.forward :124-134. */
int fvp_project_individual(const float* heat_cl, const float* cams, const int32_t* frame_set,
                           const int32_t* person_frame, const uint8_t* person_valid, const int32_t* boxes,
                           const float* fx, const float* fy, const float* fz, const int32_t* fine, int C,
                           int nP, const FvpGeom* g, float* cubes, fvp_stream_t s);

/* ---- a-15 (drop-in on a materialised cube): orthographic max projections -------------------
 * planes[p][0]=max_z -> [J][x][y], [p][1]=max_y -> [J][x][z], [p][2]=max_x -> [J][y][z];
 * planes = [nP][3][J][C][C].  Replaces joint_localization_net.py:80-81. */
int fvp_triplane_max(const float* cubes, float* planes, int nP, int J, int C, fvp_stream_t s);

/* ---- a-14 + a-15 fused (fast path): never materialises the cube ----------------------------
 * Same result as fvp_project_individual followed by fvp_triplane_max, bit for bit
 * (max is order-independent).  planes must be zero-filled by the caller beforehand
 * (hipMemsetAsync); cross-workgroup maxima use integer atomicMax on the non-negative floats.
 * persons_per_frame > 0 promises person_frame[p] == p / persons_per_frame (placement hint: the
 * workgroups of one frame are numbered onto one XCD); pass 0 if unknown.
 * A workgroup owns a compact 4 x 4 x 16 voxel block; its taps are 16-byte global loads served by the CU's L1 (no LDS
 * staging of the heatmap: DESIGN.md 4.1), LDS holds the block's plane maxima only.
 * fine_grid (may be NULL): the per-sequence cache of sampling coordinates the reference keeps
 * (project_individual.py:82-94), [nsets][V][fine0*fine1*fine2][2] as written by fvp_sample_grid on the fine
 * axes; when given, `fine` must point to the three fine-grid sizes in HOST memory and the kernel loads the
 * 8-byte coordinate of a (voxel, view) instead of recomputing the projection (same bits either way). */
int fvp_project_individual_triplane(const float* heat_cl, const float* cams, const int32_t* frame_set,
                                    const int32_t* person_frame, const uint8_t* person_valid,
                                    const int32_t* boxes, const float* fx, const float* fy, const float* fz,
                                    const int32_t* fine, int C, int nP, const FvpGeom* g, float* planes,
                                    int persons_per_frame, const float* fine_grid, fvp_stream_t s);

/* ---- a-4/a-5/a-9/a-16: conv stacks ----------------------------------------------------------
 * A stack is a list of FvpConvOp over numbered activation buffers (all NCHW fp32,
 * [planes][C][H][W]; 1-D nets use H = 1).  The same interpreter runs CenterNet
 * (cnns_2d.py:147-178), C2CNet (cnns_1d.py:112-132) and P2PNet (cnns_2d.py:115-135).
 * Convs run on the fp32 matrix cores, the kernel chosen from the layer SHAPE alone (never the batch): 3x3 layers on
 * power-of-two maps and on rows of >= 40 columns (masked tiles: CenterNet's 80- / 40-wide levels, ABI 8) as Winograd
 * F(2x2,3x3) on v_mfma_f32_16x16x4_f32 (k_conv_wino), P2PNet's 7x7 front conv on
 * 16x16x4 tiles (k_conv7), 1x1 / transposed convs register-direct (k_conv_reg), everything else as implicit GEMMs on
 * v_mfma_f32_32x32x2_f32 (k_conv_dma); the whole 1-D stack in one kernel (fvp_conv_stack_run_fused_1d).
 * BatchNorm (eval) is applied in the epilogue as  y = (acc + bias) * bn_scale + bn_shift  (no weight folding, to stay
 * close to the reference's rounding), followed by the optional residual add / ReLU. */
enum {
  FVP_OP_CONV = 0,     /* stride-1 'same' conv, KH x KW                                     */
  FVP_OP_POOL2 = 1,    /* max_pool(2,2) (2-D) or max_pool1d(2) when H == 1                  */
  FVP_OP_CONVT2 = 2    /* ConvTranspose(k2,s2) as 4 (2-D) / 2 (1-D) interleaved 1x1 GEMMs   */
};
enum {
  FVP_EPI_RELU = 1,          /* ReLU after BN (and after the residual unless RES_AFTER)     */
  FVP_EPI_RES = 2,           /* add buffer `res`                                            */
  FVP_EPI_RES_AFTER_RELU = 4 /* upsample blocks: relu(bn(x)) + skip  (cnns_2d.py:106,110)   */
};
typedef struct FvpConvOp {
  int32_t kind;
  int32_t src, dst, res;      /* activation buffer ids; res = -1 if unused                  */
  int32_t cin, cout;          /* true channel counts                                        */
  int32_t kh, kw;
  int32_t h, w;               /* INPUT spatial size                                         */
  int32_t flags;              /* FVP_EPI_*                                                  */
  int32_t w_off;              /* float offset of packed weights in `params`                 */
  int32_t e_off;              /* float offset of epilogue vectors bias|scale|shift, 3*coutp */
  int32_t cinp, coutp;        /* padded counts used by the packed layout                    */
  int32_t wino_off;           /* 0, or float offset of the Winograd-domain copy of a 3x3    */
                              /* conv's weights ([cinp][coutp][16]); when set the conv runs  */
                              /* as F(2x2,3x3) (requires even H, W a power of two)           */
  int32_t pair_off;           /* 0, or float offset of the pixel-pair copy of the weights:    */
                              /* 7x7 conv with cout <= 16 -> [cinp][7][8][32], followed by the */
                              /* k-grouped copy [max(4,ceil(cin/4))][13][4][16][4] (ABI 7,     */
                              /* fvp_conv7.h); 2-D ConvTranspose(k2,s2) ->                     */
                              /* [dy][cinp][dx*coutp+co] (fvp_conv.hip)                        */
} FvpConvOp;

/* bufs[i] = device pointer of activation buffer i (caller sized: planes*C*H*W floats).
 * plane_valid (may be NULL): planes with plane_valid[n / valid_div] == 0 are skipped. */
int fvp_conv_stack_run(const FvpConvOp* ops, int nops, const float* params, float* const* bufs, int nbufs,
                       int planes, const uint8_t* plane_valid, int valid_div, fvp_stream_t s);

/* The same interpreter for a 1-D stack (H = 1, W <= 24, cout <= 128: C2CNet) fused into ONE
 * kernel, one workgroup per plane, activations resident in LDS.  in = [planes][cin][W] (buffer
 * ops[0].src), out = [planes][cout][W'] of the last op.  Results are identical to
 * fvp_conv_stack_run (same accumulation order). */
int fvp_conv_stack_run_fused_1d(const FvpConvOp* ops, int nops, const float* params, const float* in,
                                float* out, int planes, fvp_stream_t s);

/* Pack one conv's parameters from the reference's state_dict tensors (device copies):
 * weight [cout][cin][kh][kw] (or [cin][cout][kh][kw] when transposed) -> [tap][cinp][coutp];
 * bias / BN vectors -> bias|scale|shift with scale = gamma/sqrt(var+eps), shift = beta - mean*scale
 * (scale = 1, shift = 0 when bn_* are NULL). */
int fvp_pack_conv(const float* weight, const float* bias, const float* bn_gamma, const float* bn_beta,
                  const float* bn_mean, const float* bn_var, float eps, int transposed, const FvpConvOp* op,
                  float* params, fvp_stream_t s);

/* ---- a-6/a-7: NMS + top-k (bit-exact indices) -------------------------------------------------
 * hm2d [B][X][Y].  keep = (x == maxpool3x3(x)) ? x : 0 ; top-N by (value desc, flat index asc).
 * vals [B][N] fp32, idx [B][N][2] int64 = (flat / X, flat % X) -- the reference divides by
 * shape[1] = X (core/proposal.py:16-17), flat [B][N] int64.  Replaces core/proposal.py:13-33.
 * One workgroup per frame with the map in LDS: FVP_ELIMIT when X * Y * 4 + 128 bytes exceed the CU's 160 KB
 * (X * Y > 40 928, e.g. beyond 200 x 200); maps up to 128 x 128 take the register-resident fast path. */
int fvp_nms_topk(const float* hm2d, int B, int X, int Y, int N, float* vals, int64_t* idx, int64_t* flat,
                 fvp_stream_t s);

/* ---- a-8: gathers at the top-k cells ------------------------------------------------------------
 * bbox_map [B][2][X][Y] -> bbox_flat [B][X*Y][2] (the 4th output of HumanDetectionNet.forward,
 * may be NULL) and match_bbox [B][N][2]; cubes [B][J][X][Y][Z] -> feat1d [B*N][J][Z]
 * (cubes and feat1d may both be NULL: the columns then come from fvp_project_columns).
 * Replaces human_detection_net.py:88-93. */
int fvp_gather_proposals(const float* bbox_map, const float* cubes, const int64_t* flat, int B, int J, int X,
                         int Y, int Z, int N, float* bbox_flat, float* match_bbox, float* feat1d,
                         fvp_stream_t s);

/* ---- a-10/a-11: z arg-max, confidence product, proposal packing ---------------------------------
 * hm1d [B*N][Z]; idx2d from fvp_nms_topk.  topk_index [B][N][3] int64 (may be NULL);
 * centers [B][N][7] = (x,y,z mm = idx*scale + bias as fp32 mul then add, no FMA), valid-1,
 * conf, bbox_w, bbox_h.  sb = scale[3], bias[3].  valid [B][N] uint8 (may be NULL) = centers[...,3] >= 0,
 * the `mask` faster_voxelpose.py:45 hands to the joint localisation net (ABI 8: written here instead of by a
 * separate comparison launch).  Replaces human_detection_net.py:95-102 and ProposalLayer.forward :44-65
 * (eval branch). */
int fvp_proposals(const float* hm1d, const float* conf2d, const int64_t* idx2d, const float* match_bbox,
                  const float* sb, float min_score, int B, int N, int Z, int64_t* topk_index, float* centers,
                  uint8_t* valid, fvp_stream_t s);

/* ---- a-11 standalone: ProposalLayer.forward, eval branch (human_detection_net.py:44-65) -----------
 * topk_index [B][N][3] int64 (voxel indices x, y, z), topk_confs [B][N], match_bbox [B][N][2];
 * centers [B][N][7] = (idx.float() * scale + bias as fp32 mul then add, no FMA), (conf > min_score) - 1,
 * conf, bbox_w, bbox_h.  sb = scale[3], bias[3].  The forward path uses the fused fvp_proposals; this is
 * the module-level drop-in for callers that invoke the layer on its own. */
int fvp_proposal_layer(const int64_t* topk_index, const float* topk_confs, const float* match_bbox, const float* sb,
                       float min_score, int B, int N, float* centers, fvp_stream_t s);

/* ---- a-17/a-18 first half: soft-argmax + WeightNet per (person, plane, joint) map ----------------
 * feat [nP][3][J][C][C] (P2PNet output).  For each map: softmax(beta*x) over C*C cells,
 * expectation of center_grid[plane] ([3][C*C][2]) accumulated in fp64, max probability;
 * WeightNet (conv 1->F k3 + BN + maxpool2 + ReLU + global avg + MLP F->Hd->1 + sigmoid,
 * lib/models/weight_net.py:69-80) from wn = packed WeightNet parameters (see fvp_pack_weightnet).
 * pose2d [nP][3][J][2] (offset NOT yet added), pmax [nP][3][J], wgt [nP][3][J]. */
int fvp_softargmax_weightnet(const float* feat, const float* center_grid, const float* wn, float beta,
                             int nP, int J, int C, int F, int Hd, const uint8_t* person_valid, float* pose2d,
                             float* pmax, float* wgt, fvp_stream_t s);
int fvp_pack_weightnet(const float* conv_w, const float* conv_b, const float* bn_gamma, const float* bn_beta,
                       const float* bn_mean, const float* bn_var, float eps, const float* fc1_w,
                       const float* fc1_b, const float* fc2_w, const float* fc2_b, int F, int Hd, float* wn,
                       fvp_stream_t s);

/* ---- a-18 second half / a-19: offsets, fusion, scatter-back ---------------------------------------
 * Adds offset per plane (joint_localization_net.py:87-90), normalises the weight pairs and
 * blends (:44-62), conf = mean over (plane, joint) of pmax (:27-28); writes rows of
 * fused_poses [B*N][J][5] (x,y,z, valid flag, conf), plane_poses [3][B*N][J][2] and
 * proposal_centers[..][4] = conf for valid persons (:96-98, faster_voxelpose.py:102-103).
 * Invalid persons get xyz = 0 and keep their HDN confidence. */
int fvp_fuse_poses(const float* pose2d, const float* pmax, const float* wgt, const float* offset,
                   const uint8_t* person_valid, int nP, int J, float* centers, float* fused_poses,
                   float* plane_poses, fvp_stream_t s);

/* ---- "next" row f-2: input heatmaps rasterised from 2-D detections ------------------------------------
 * joints [nimg][P][J][2] float64 in NETWORK-image pixels (already through the resize affine, as
 * JointsDataset.py:149-151 leaves them), num_people [nimg] <= P.  Writes Gaussian heatmaps
 * [nimg][J][H][W] (heat_nchw, the reference layout) and / or the channels-last staging copy
 * [nimg][H*W][JP] the projection kernels read (heat_cl; either may be NULL).  Replaces
 * JointsDataset.generate_input_heatmap (lib/dataset/JointsDataset.py:271-338, eval branch) and
 * compute_human_scale (:197-203); float64 scalar arithmetic like numpy, one rounding to fp32. */
int fvp_rasterise_heatmaps(const double* joints, const int32_t* num_people, int nimg, int P, int J, int W, int H,
                           double feat_stride_x, double feat_stride_y, double sigma, float* heat_nchw,
                           float* heat_cl, int JP, fvp_stream_t s);

/* ---- "next" row f-1: Pose-ResNet backbone in bf16 (lib/models/resnet.py:98-215) ------------------------
 * Activations are NHWC bf16 (uint16 storage; the image input is padded to 8 channels), every conv /
 * ConvTranspose(k4,s2,p1) is an implicit GEMM on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, eval
 * BatchNorm folded into a per-cout scale / shift, residual add and ReLU in the epilogue.  The op flagged
 * FVP_BB_OUT_HEAT (final_layer) writes fp32 heatmaps: channels-last [N][H*W][heat_jp] (what the projection
 * kernels read) and / or NCHW [N][J][H][W] (the reference's layout). */
enum { FVP_BB_CONV = 0, FVP_BB_MAXPOOL = 1, FVP_BB_DECONV = 2 };
enum { FVP_BB_OUT_HEAT = 8,      /* with FVP_EPI_RELU = 1 in `flags` */
       FVP_BB_STEM = 16,         /* the 7x7 / stride-2 stem on a <= 4-channel image: pixel-pair form (fvp_backbone.hip) */
       FVP_BB_CFG_SHIFT = 8 };   /* flags bits 8-9: tile configuration chosen by fvp_bb_tune (0 = built-in heuristic) */
typedef struct FvpBbOp {
  int32_t kind;
  int32_t src, dst, res;      /* activation buffer ids (dst = -1 for the heatmap op, res = -1 if unused) */
  int32_t cin, cinp;          /* true / stored input channels (cinp: power of two >= 8)                 */
  int32_t cout, coutp, cbuf;  /* true couts, couts of the packed weights (multiple of 64), channels of dst */
  int32_t kh, kw, stride, pad;
  int32_t h, w, oh, ow;       /* input and output spatial size                                          */
  int32_t flags;
  int32_t w_off;              /* bf16 element offset of the packed weights in wblob                     */
  int32_t e_off;              /* float offset of scale | shift (2 * coutp) in eblob                      */
} FvpBbOp;
/* images [N][C<=4][H][W] fp32 (W even) -> NHWC bf16 with 4 channels per pixel = [N][H][W/2][8] pixel pairs */
int fvp_bb_input(const float* images, uint16_t* nhwc8, int N, int C, int H, int W, fvp_stream_t s);
/* state_dict tensors of one conv (+ its BatchNorm, may be NULL) -> packed bf16 weights [cls][coutp][taps*cinp]
 * and fp32 scale | shift (resnet.py conv / bn / deconv / final_layer modules) */
int fvp_bb_pack(const float* weight, const float* bias, const float* bn_gamma, const float* bn_beta,
                const float* bn_mean, const float* bn_var, float eps, const FvpBbOp* op, uint16_t* wblob,
                float* eblob, fvp_stream_t s);
/* run the op list on N images; bufs[i] = NHWC bf16 activation buffer i.  The first 64 floats of eblob must be
 * zero (e_off >= 64): the LDS-DMA of the large-tile kernel reads them for padding.
 * Op patterns the list holds in sequence run as ONE kernel each, with the bits of the op-by-op launches (ABI 8):
 * stem conv (+ bn, ReLU) followed by its max-pool (resnet.py:103-106, :185-188), and a 64-plane bottleneck
 * [1x1 downsample,] 1x1 -> 3x3 -> 1x1 + residual (resnet.py:57-95) on one map - provided nothing else in the list
 * reads the intermediate buffers, which then stay unwritten.  The 1x1 heatmap layer behind the last transposed
 * conv is applied in that conv's epilogue. */
int fvp_bb_run(const FvpBbOp* ops, int nops, const uint16_t* wblob, const float* eblob, void* const* bufs, int nbufs,
               int N, float* heat_cl, int heat_jp, float* heat_nchw, fvp_stream_t s);
/* optional, once per (op list, N): times every conv op with each tile configuration of the large-tile kernel and
 * records the fastest in its flags (the configurations compute identical bits).  Synchronises `s`; the activation
 * buffers are used as scratch. */
int fvp_bb_tune(FvpBbOp* ops, int nops, const uint16_t* wblob, const float* eblob, void* const* bufs, int nbufs, int N,
                fvp_stream_t s);

/* ---- measurement hooks (bench.py roofline leg) -----------------------------------------------------
 * fvp_prof_enable(1): kernel classes are bracketed by hipEvents on the stream they are launched
 * on -- per launch for the projection / soft-argmax / small kernels, one pair per
 * fvp_conv_stack_run for the conv class (launches = convs in the stack), so that a ~100-launch
 * step is not perturbed.  fvp_prof_enable(2): one pair per conv LAUNCH instead, the Winograd 3x3
 * launches (the dominant kernel) in their own class FVP_K_CONV_WINO, all others in FVP_K_CONV.  Winograd launches with
 * fewer work units than the chip has workgroup slots (CenterNet's levels since round 6, everything at B = 1) are
 * launch-latency-bound, not matrix-core-bound: they go to FVP_K_CONV_WINO_SMALL so that the roofline of the
 * chip-filling launches stays what it describes.
 * fvp_prof_read synchronises the events and returns accumulated milliseconds, launch count and
 * algorithmic FLOPs since the last reset. */
enum { FVP_K_PROJECT_WHOLE = 0, FVP_K_PROJECT_TRIPLANE = 1, FVP_K_CONV = 2, FVP_K_SOFTARGMAX = 3,
       FVP_K_OTHER = 4, FVP_K_CONV_WINO = 5, FVP_K_BACKBONE = 6, FVP_K_CONV_WINO_SMALL = 7, FVP_K_COUNT = 8 };
int fvp_prof_enable(int on);
int fvp_prof_read(int cls, double* ms, int64_t* launches, double* flops);
int fvp_prof_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* FVP_H_ */
