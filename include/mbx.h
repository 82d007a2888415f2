/* libmbx -- C-ABI boundary of the MI355X-native Multibox hot path.
 *
 * The reference (gvanhorn38/multibox, TF-0.11 Python) has no FFI of its own; its one
 * host-callback seam is tf.py_func(compute_assignments, ...) at loss.py:81-82 and the
 * numpy post-processing loop at detect.py:408-443.  Every entry point below names the
 * reference interface (file:line) it replaces.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; device pointers unless marked HOST;
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no allocation, no
 *     synchronisation, no global state: safe to capture into a hipGraph;
 *   - return 0 (MBX_OK) or a negative mbx_status; never throw across the boundary;
 *   - scratch memory comes from the caller; *_workspace_bytes() says how much.
 */
#ifndef MBX_H
#define MBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mbx_stream_t; /* hipStream_t */

typedef enum {
  MBX_OK = 0,
  MBX_ERR_INVALID_ARG = -1,
  MBX_ERR_UNSUPPORTED = -2,
  MBX_ERR_LAUNCH = -3,
  MBX_ERR_WORKSPACE = -4
} mbx_status;

int mbx_version(void);
const char* mbx_status_string(int status);

/* ---------------------------------------------------------------- priors (HOST)
 * Replaces priors.generate_priors (priors.py:185-314).  float64, bit-exact.
 * `grids` generalises the hard-coded [8,6,4,3,2,1] (priors.py:196); a grid of 1 gets
 * the single aspect-1 box (priors.py:206-258).  out = [rows,4] x1,y1,x2,y2.          */
int mbx_priors_count(int k, const int* grids, int n_grids);

/* CRC-32C (Castagnoli) of a host buffer, chained through `crc` (0 first): the checksum of TFRecord frames
 * (inputs.py:225-247 reads them with tf.TFRecordReader) and of TF checkpoint table blocks (train.py:15-90). Host code. */
uint32_t mbx_crc32c(const void* data, uint64_t n, uint32_t crc);
int mbx_generate_priors(const double* aspect_ratios, int k, double min_scale, double max_scale,
                        int restrict_to_image_bounds, const int* grids, int n_grids,
                        double* out /*HOST [rows,4]*/);

/* ------------------------------------------------------- decode + confidences (K12)
 * loss.py:67-74: decoded = raw_locs + tile(priors); conf = sigmoid(logits) + eps_add.
 * (model.py:322 applies the sigmoid; eps_add = 1e-10 for the loss, 0 for detect.)
 * Any output pointer may be NULL.                                                     */
int mbx_decode_conf(const float* raw_locs /*[B,P,4]*/, const float* logits /*[B,P]*/,
                    const float* priors /*[P,4]*/, int B, int P, float eps_add,
                    float* decoded /*[B,P,4]*/, float* conf /*[B,P]*/, mbx_stream_t stream);

/* ------------------------------------------------------------------ matching (A6)
 * Replaces the tf.py_func(compute_assignments) callback (loss.py:8-53, 81-82).
 * Inputs exactly as the callback receives them: prior-decoded locations, confidences
 * with 1e-10 already added, zero-padded gt, counts.  Per image: cost
 * C[p,j] = (alpha/2)*||l_p - g_j||^2 - log c_p + log(1-c_p) built in float32 in the
 * reference's operation order (loss.py:21-35), solved as a rectangular linear sum
 * assignment in float64 (shortest augmenting path; scipy's linear_sum_assignment,
 * loss.py:40).  match[b,p] = gt index or -1; the reference's 0/1 partition and
 * row-ordered stacked_gt (loss.py:44-48) follow from it.
 * status[b]: 0 ok, 1 n_gt > P (infeasible), 2 non-finite cost (scipy raises there).    */
size_t mbx_match_workspace_bytes(int B, int P, int G);
int mbx_match(const float* decoded /*[B,P,4]*/, const float* conf /*[B,P]*/,
              const float* gt /*[B,G,4]*/, const int32_t* n_gt /*[B]*/, float alpha,
              int B, int P, int G, int32_t* match /*[B,P]*/, int32_t* status /*[B]*/,
              void* workspace, size_t workspace_bytes, mbx_stream_t stream);

/* ---------------------------------------------------------------- loss fwd+bwd (A7)
 * loss.py:88-101 given the matching: loc_loss = alpha * 1/2 sum (decoded-gt)^2 over
 * matched rows; conf_loss = -sum log(c_matched) - sum log(1 - c_unmatched + 1e-10),
 * c = sigmoid(logit) + 1e-10.  Also the gradients w.r.t. raw_locs and the logits (no
 * gradient through the matching: py_func, loss.py:82).  loss2 = {loc_loss, conf_loss}.
 * grad_scale multiplies both gradients (1 for the reference's batch-sum loss).
 * conf_is_logit = 1: `conf_in` holds logits (gradient goes through the sigmoid of
 * model.py:322); 0: `conf_in` holds sigmoid outputs as loss.add_loss receives them
 * (d_logits then is the gradient w.r.t. those confidences).                           */
size_t mbx_loss_workspace_bytes(int B);
int mbx_loss_fwd_bwd(const float* decoded /*[B,P,4]*/, const float* conf_in /*[B,P]*/,
                     int conf_is_logit, const float* gt /*[B,G,4]*/,
                     const int32_t* match /*[B,P]*/, float alpha, float grad_scale, int B, int P,
                     int G, float* loss2 /*[2]*/,
                     float* d_raw_locs /*[B,P,4] or NULL*/, float* d_logits /*[B,P] or NULL*/,
                     void* workspace, size_t workspace_bytes, mbx_stream_t stream);

/* ------------------------------------------------- detect post-processing (A9-A12)
 * Replaces the per-patch numpy loop detect.py:408-436: decode + clip [0,1]
 * (412-413), filter_proposals (74-104, strict inequalities), sort by confidence
 * descending and keep max_to_keep (423-427), convert_proposals to image coordinates in
 * float64 incl. the flip (106-131).  Ties in confidence: higher prediction index first
 * (the reference's order among ties is undefined).  `conf` may hold ANY float (the sort key is an order-
 * preserving image of the float bits): negative values, values >= 1 and infinities order as numpy's
 * argsort does; a NaN sorts first, as argsort(...)[::-1] (detect.py:423) places it.            */
typedef struct {
  int32_t offset_y, offset_x; /* batched_offsets  (detect.py:190-281) */
  int32_t patch_h, patch_w;   /* batched_dims */
  int32_t image_h, image_w;   /* batched_heights_widths */
  int32_t is_flipped;
  int32_t max_to_keep;
  float restrictions[4];      /* x1,y1,x2,y2 (detect.py:50-54) */
} mbx_patch_meta;

int mbx_decode_filter_topk(const float* raw_locs /*[B,P,4]*/, const float* conf /*[B,P]*/,
                           const float* priors /*[P,4]*/, const mbx_patch_meta* meta /*[B]*/,
                           int B, int P, int k_max, double* out_boxes /*[B,k_max,4]*/,
                           float* out_scores /*[B,k_max]*/, int32_t* out_index /*[B,k_max]*/,
                           int32_t* out_count /*[B]*/, mbx_stream_t stream);

/* OPTIONAL greedy non-maximum suppression per patch on the output of mbx_decode_filter_topk, in place (row N1: BASELINE's
 * north_star names an NMS stage; the reference has none -- detect.py:408-443 keeps the top max_to_keep boxes -- so it is
 * off by default and outside the parity path).  Boxes are taken in their stored (score-descending) order; box i is
 * dropped iff IoU(i, j) > iou_threshold for an earlier KEPT box j; survivors are compacted to the front of each row of
 * out_boxes / out_scores / out_index and out_count[b] becomes their number.  IoU in float64, k_max <= 1024.      */
int mbx_nms(double* boxes /*[B,k_max,4] x1,y1,x2,y2*/, float* scores /*[B,k_max]*/, int32_t* index /*[B,k_max]*/,
            int32_t* count /*[B], in/out*/, int B, int k_max, double iou_threshold, mbx_stream_t stream);

/* ------------------------------------------------------------ convolution stack (A2-A4)
 * Replaces slim.conv2d (+ batch_norm + relu) of model.py:6-324 and its TF gradients
 * (train.py:263).  Activations are NHWC bf16 *views*: element (n,h,w,c) of a tensor lives at
 * base[n*img_stride + (h*W + w)*ld + c], so a branch can read/write a channel slice of a
 * wider concat buffer (tf.concat(3, ...) of model.py:18,38,58,138,159,182 costs nothing).
 * Filters are bf16 [C_out][R][S][C_in] (KRSC), contiguous.  Accumulation is fp32 on MFMA
 * (v_mfma_f32_16x16x32_bf16).  C_in, ld and slice offsets must be multiples of 8.
 *
 * mbx_conv_desc.transposed = 1 computes the data gradient: the same kernel run on dy with
 * the spatially flipped, channel-transposed filter ([C_in][R][S][C_out], see
 * mbx_filter_prepare) and an input dilated by `stride`.                                 */
typedef enum {
  MBX_EPI_STORE = 0,    /* y = acc [* rscale] [+ y if accumulate] [* (skip > 0): relu backward mask]
                           (bf16; pre-BN activations, gradients)                           */
  MBX_EPI_AFFINE = 1,   /* y = act(acc*scale[c] + shift[c])       (frozen / folded batch norm) */
  MBX_EPI_RESIDUAL = 2, /* y = act(skip + rscale*(acc + shift[c]))           (model.py:19-23) */
  MBX_EPI_STORE_F32 = 3 /* y = acc as float32            (head outputs, model.py:213-293)     */
} mbx_epilogue;

/* BATCH-NORM BACKWARD STATISTICS FROM THE DATA GRADIENT THAT WRITES THE ACTIVATION GRADIENT (round 4).  The backward pass of
 * slim.batch_norm + relu (train.py:94-99) needs, per channel, sum g and sum g xhat over all pixels before it can write one
 * element -- with g = da (a > 0): a grid-wide dependency that cost a grid barrier per layer (mbx_bn_bwd_onepass) or three
 * launches.  The convolution whose data gradient WRITES da already holds every element of it in registers: with this table
 * attached to its descriptor (mbx_conv_desc.bn_bwd_stats) its epilogue also reads y at the same positions and ADDS
 *   sum g  and  sum g y,   g = (y > relu_thr[c]) ? bf16(da) : 0
 * (float32 atomics) into row (tile index mod rows_mod) of stats[i] = [rows_mod][stats_ld[i]][2], ZERO at launch.  Output
 * channels [c_begin[i], c_begin[i+1]) of the launch belong to entry i (at most 4: the gradient of a concat buffer slice
 * feeds several layers): y[i] = that layer's pre-BN output [M, ld_y[i]] at the entry's first channel, relu_thr[i] / stats[i]
 * likewise.  c_begin ascending from 0 in multiples of 32.  mbx_bn_bwd_apply_rows then is ONE streaming launch.
 * (sum g xhat = rstd (sum g y - mean sum g); the mask y > mean - beta / rstd is the forward's (y - mean) rstd + beta > 0.)
 * Plain bf16 STORE data gradients only (no accumulate, no mask, stride 1); anything else: MBX_ERR_INVALID_ARG / UNSUPPORTED. */
typedef struct {
  int32_t n, rows_mod;
  int32_t c_begin[4];
  const void* y[4]; int32_t ld_y[4];
  const float* relu_thr[4];
  float* stats[4]; int32_t stats_ld[4];
} mbx_bn_bwd_stats;

/* THE BATCH-NORM APPLY OF A TRAINING-MODE CONVOLUTION AS THE TAIL OF ITS OWN LAUNCH (round 6).  slim.conv2d in training mode
 * (train.py:94-105: convolution, batch statistics, normalise + beta, relu) was two launches: the convolution, adding its
 * tile sums into the fixed-point statistics rows (stats_rows_mod > 0), and mbx_bn_apply_fused_mapped.  With this table
 * attached (mbx_conv_desc.bn_apply) the convolution's workgroups, when their last tile is stored, meet at a ONE-SHOT GRID
 * BARRIER, reduce the rows of every channel themselves and normalise THE TILES THEY WROTE (read back from their own CU's L2):
 * one launch, the same mean / rstd / relu threshold / moving statistics / activations bit for bit.
 *   barrier: DEVICE, MBX_GRID_BARRIER_BYTES, 128-byte aligned, ZERO at launch, used by one launch per clearing.
 *   The launch takes at most one workgroup per CU (max_workgroups caps it further); if its workgroups are NOT co-resident
 *   (CUs held by another stream's kernels) the barrier times out: those workgroups write NaN activations and add 1 to
 *   *step_poison (word [0] of the step control block: the optimiser then skips the step on every rank) -- the caller falls
 *   back to the two-launch form.
 * Persistent / one-tile-per-workgroup launches only (tile_config 33..39, 97, 98), bf16 STORE with statistics, y rows
 * contiguous per image (y_img_stride = H_out W_out ldy), not a member of mbx_conv_pair; anything else: MBX_ERR_UNSUPPORTED. */
#define MBX_GRID_BARRIER_BYTES 6400
typedef struct {
  void* barrier;
  void* a; int32_t ld_a;                /* activation: a[m * ld_a + c], m = img * H_out * W_out + pixel (bf16)      */
  const float* beta;                    /* [C_out]                                                                   */
  float* mean; float* rstd;             /* [C_out] out: batch mean, 1 / sqrt(var + eps)                              */
  float* moving_mean; float* moving_var;/* may be NULL; decay < 0: STORE mode (batch mean / biased variance), else
                                           moving -= (1 - decay) (moving - batch)                                    */
  float* relu_thr;                      /* may be NULL: mean - beta / rstd (relu) or -inf                            */
  int32_t relu; float eps, decay;
  float* step_poison;                   /* may be NULL                                                               */
} mbx_bn_apply_desc;

/* THE BATCH-NORM BACKWARD OF THE LAYERS A DATA GRADIENT FEEDS, AS THE TAIL OF THAT DATA-GRADIENT LAUNCH (round 6).  The
 * convolution whose data gradient WRITES the activation gradient da of batch-norm layers (train.py:94-99 on the way back,
 * train.py:263) finishes their backward itself: when a workgroup's last tile is stored it sweeps THE TILES IT WROTE (da back
 * from its own CU's L2, y from memory) for its share of  sum g  and  sum g xhat  (g = da where the activation was positive:
 * xhat + beta > 0 recomputed from y), adds them into the layers' accumulators (float atomics), meets the other workgroups at
 * a ONE-SHOT GRID BARRIER, reads the totals and sweeps its tiles again to write
 *   dy = rstd (g - mean g - xhat mean g xhat),   dbeta += sum g
 * -- what mbx_bn_bwd_onepass computed in a launch of its own.  Output channels [c_begin[i], c_begin[i+1]) of the launch belong
 * to entry i (at most 4: the gradient of a concat-buffer slice feeds several layers); y / dy / mean / rstd / beta / dbeta /
 * acc are that layer's, AT THE ENTRY'S FIRST CHANNEL; acc = [MBX_BN_BWD_SLOTS][2][acc_ld] float32, ZERO at launch; c_begin
 * ascending from 0 in multiples of 8; relu[i]: the layer has a relu behind its batch norm.  Every channel of da must be
 * written by this launch alone (plain bf16 STORE, no accumulate / mask / sign bits / statistics, stride 1, rows of da
 * contiguous per image).  barrier / step_poison / residency / time-out: as mbx_bn_apply_desc.  tile_config 33..39, 97, 98
 * only; anything else: MBX_ERR_UNSUPPORTED.  Same mathematics as mbx_bn_bwd_onepass; like it, the float atomics make the
 * last bits depend on the order of arrival.                                                                              */
#define MBX_BN_BWD_SLOTS 8
typedef struct {
  void* barrier;
  int32_t n;
  int32_t c_begin[4];
  const void* y[4]; int32_t ld_y[4];
  void* dy[4]; int32_t ld_dy[4];
  const float* mean[4]; const float* rstd[4]; const float* beta[4];
  float* dbeta[4];
  float* acc[4]; int32_t acc_ld[4];
  int32_t relu[4];
  float* step_poison;
} mbx_bn_bwd_fused;

typedef struct {
  /* input view */
  const void* x; int64_t x_img_stride; int32_t ldx;
  int32_t N, H_in, W_in, C_in;
  /* filter */
  const void* w; int32_t C_out, R, S;
  /* geometry */
  int32_t stride, transposed, pad_t, pad_l, H_out, W_out;
  /* output view */
  void* y; int64_t y_img_stride; int32_t ldy;
  /* epilogue */
  int32_t epilogue, relu, accumulate;     /* accumulate: y += result (bf16 STORE only)        */
  const float* scale; const float* shift; /* per output channel, may be NULL                 */
  const void* skip; int64_t skip_img_stride; int32_t ld_skip; float rscale;
  float* stats_partial; /* [mbx_conv_stats_rows()][C_out][2] partial sum / sum-of-squares of the stored
                           y (bf16-rounded) for batch-norm statistics, or NULL              */
  int32_t tile_config;  /* 0: the library picks the tile; n > 0: use tile configuration n-1 (0..MBX_CONV_TILE_CONFIGS-1)
                           -- for callers that time the candidates on their own shapes.  Results do not depend on
                           it (same K order), only mbx_conv_stats_rows() does.  For mbx_conv_wgrad*: 0 default,
                           1..6 = {8 waves x 256 blocks, 4 waves x 512, 8 x 192, 4 x 384, 8 x 128, 8 x 224}, 7 / 8 =
                           the narrow tile (64 output channels per block): 8 waves x 256 / 192, 9 / 10 = 4 waves x 512 / 768;
                           11 = un-split (one block per output tile: every dw element is added to once, so the
                           result is bit-reproducible from run to run -- the MBX_DETERMINISTIC debug mode). */
  /* accumulate != 0: y = result + OLD, where OLD is read from acc_src (same shape as y) if it is not NULL, else
     from y itself.  Lets the residual trunk gradient of model.py:21 be built out of place, G[i-1] = G[i] + dgrad,
     so that G[i] survives for the deferred weight gradient of block i's 1x1 "up" convolution.          */
  const void* acc_src; int64_t acc_img_stride; int32_t ld_acc;
  /* Persistent launches (tile_config > 32: conv_igemm5_kernel, one workgroup per CU that walks several tiles): DEVICE
     int32 that is ZERO at launch, or NULL.  With it, a workgroup takes its first tile by position and every further
     tile from this counter (one atomic per tile, fetched two tiles ahead), so a workgroup that starts late -- its CU
     held by another stream's kernels, e.g. RCCL in a data-parallel run -- simply takes fewer tiles; with NULL the tiles
     are dealt statically (first, first + grid, ...) and a late workgroup finishes its share late.  The launch leaves
     the counter advanced: clear it before the next launch that uses it (one fill per step covers a whole table of
     them).  Results are identical either way.                                                                     */
  int32_t* work_counter;
  /* Persistent launches only (tile_config > 32): 0 = one workgroup per CU; n > 0 = at most n workgroups, which leaves
     CUs to a kernel running beside this one on another stream (the capped grouped weight gradient of the previous
     backward segment, mbx_conv_wgrad_grouped_capped; RCCL).  Results do not depend on it.                          */
  int32_t max_workgroups;
  /* tile_config 128 + S (S = 2..32): SPLIT-K for long-K convolutions with few output tiles (forward, bf16 store with or
     without statistics): S slices of the K range per 128 x 64 tile as float32 partial tiles in this workspace
     (mbx_conv_splitk_workspace_bytes(), 16-byte aligned), then one launch that adds the slices in slice order
     (deterministic), rounds, stores y and writes the statistics partials (a row per 16 pixels).  The float32 sum is
     grouped differently from the one-pass kernels: results agree with them to 1 bf16 ulp, not bit for bit.          */
  void* splitk_ws; int64_t splitk_ws_bytes;
  /* stats_rows_mod = R > 0 (round 4): the statistics of a tile are ADDED into row (tile index mod R) of a table
     [R][stats_ld][2] of INT64 that stats_partial then points at (16 bytes per channel and row) and that must be ZERO at
     launch -- R rows whatever the tile shape (mbx_conv_stats_rows() = R), few enough for the consumer to reduce them itself
     (mbx_bn_apply_fused_mapped: no finalize launch), many enough that the adders of one address stay few.  The sums are
     added as 64-bit integers in fixed point (units of 2^-20, resolution 1e-6): integer addition is associative, so the table
     does not depend on the order in which the tiles arrive -- bit-reproducible statistics.  RANGE (round 6): the table is
     exact while a channel's GRAND TOTAL stays in range: |sum y| and sum y^2 over all N H W pixels below 2^42 = 4.4e12 (2^62
     in fixed point; e.g. rms |y| < 1.5e4 on a 64 x 17 x 17 map, < 1.8e3 on 64 x 147 x 147).  Every adder (a pixel tile, a
     workgroup, an image -- the launch knows how many add into one channel) is held to its share 2^42 / adders, so that
     neither a row nor the consumer's sum over the rows can wrap; an adder whose sums are not finite or not below its share
     POISONS the channel -- the sum-of-squares word is forced to INT64_MIN by a signed atomic min, and since the legitimate
     adds are non-negative and total less than 2^62 it stays negative whatever arrives before or after -- and
     mbx_bn_apply_fused_mapped reports NaN mean / rstd for a channel with a negative word or total: out-of-range activations
     end in NaN (as the float32 rows' inf - inf would), never in finite garbage.
     stats_ld (0: C_out): channels per row -- sibling convolutions of a batch-norm group add into channel slices of one
     table (stats_partial then points at the member's first channel).                                              */
  int32_t stats_rows_mod, stats_ld;
  const mbx_bn_bwd_stats* bn_bwd_stats;   /* HOST, may be NULL: see above */
  /* ReLU SIGN BITS of a residual block output (round 4), or NULL: one bit per element, byte [m * ld_bits + c / 8] bit
     (c & 7) = (y[m][c] > 0), m = img * H_out * W_out + pixel, c = channel within this launch's C_out; ld_bits bytes per
     pixel, a multiple of 4 and >= 4 ceil(C_out / 32); the table 4-byte aligned.  MBX_EPI_RESIDUAL with relu: the launch
     WRITES them beside y, four bytes (32 channels) at a time: bytes [0, 4 ceil(C_out / 32)) of every pixel row, zeros for
     channels past C_out.  MBX_EPI_STORE: the launch READS them as the relu-backward mask -- y = (acc [* rscale] [+ OLD]) where the
     bit is set, else 0 -- in place of the `skip` tensor (which must then be NULL): 1/16 of its bytes on a launch that is
     bound by the three tensors its epilogue streams (model.py:19-23 on the way back; train.py:263).  Same results as
     the `skip` form (a NaN in the block output counts as positive here, as not positive there).  Not with the
     stride-2 data gradient, statistics, split-K, the direct launches (96 / 97) or mbx_conv_pair: MBX_ERR_UNSUPPORTED. */
  void* relu_bits; int32_t ld_bits;
  const mbx_bn_apply_desc* bn_apply;      /* HOST, may be NULL: see above (round 6) */
  const mbx_bn_bwd_fused* bn_bwd;         /* HOST, may be NULL: see above (round 6) */
} mbx_conv_desc;
#define MBX_CONV_TILE_CONFIGS 14
/* tile_config 33..39: the persistent igemm5 launch (128x64, 128x128, 192x128, 256x128, 256x64, 128x192, 128x256 tiles; 128x192
   without the accumulate + mask epilogue); 65: the persistent
   POINTWISE launch with the filter panel resident in LDS (1x1, unit stride, unpadded, 64 < K <= 384, no statistics; its
   work_counter, if given, is an array of one zeroed int32 per 128-channel column tile, at most 32).  A configuration that
   does not apply returns MBX_ERR_UNSUPPORTED.
   tile_config 96: the DIRECT 3x3 launch (stride 1, forward or data gradient, C_in 32 / 64, C_out <= 64; bf16 store with or
   without statistics, or the affine epilogue) for few channels on large maps: a persistent workgroup per CU stages a pixel
   patch with its halo once and multiplies the nine taps out of LDS instead of gathering the input nine times -- and, under
   the same number, the network's first layer (3x3 / stride 2, C_in 8 = the packed RGB input, C_out <= 32; forward only).
   tile_config 97: the same scheme with WHOLE-WIDTH tiles for narrow maps (8..64 wide: block35's 35 x 35 layers; C_in 32 /
   48 / 64, C_out <= 64).  Same K order and MFMA grouping as the implicit-GEMM tiles: bit-identical outputs;
   mbx_conv_stats_rows() = one row per workgroup.
   tile_config 98 (round 5): the RESIDENT-IMAGE launch for stride-1, same-size convolutions with a one-dimensional multi-tap
   filter on small maps with many channels -- 1x7 / 7x1 on maps of 65..289 pixels with C_in 128 / 160 / 192 (block17,
   model.py:33-37) and 1x3 / 3x1 on 8 x 8 maps with C_in 192 / 224 / 256 (block8, model.py:53-57), forward or data gradient,
   bf16 store with or without statistics or the affine epilogue: a tile = one whole image (staged in LDS once) x a quarter of
   the output channels, the filter streamed per tap.  Bit-identical outputs; mbx_conv_stats_rows() = N (a row per image).
   tile_config 99 (round 5): the PIXEL-RESIDENT POINTWISE launch (1x1, unit stride, unpadded, C_in 96 / 128 / 320 / 384 / 448)
   for the residual epilogue (+ relu, + sign bits) or a store masked by relu sign bits (with or without an accumulate
   source): a 160- / 128-pixel tile resident in LDS, the filter streamed per 128 output channels.  Bit-identical outputs; no
   statistics.  (Built, tested and level with the persistent tiles at BATCH_SIZE 64: not chosen by the engine's rules.) */

int mbx_conv_stats_rows(const mbx_conv_desc* desc /*HOST*/); /* rows of stats_partial */
int mbx_conv(const mbx_conv_desc* desc /*HOST*/, mbx_stream_t stream);
/* Every check of mbx_conv for this descriptor -- arguments, ranges, whether desc->tile_config applies to the shape and
 * epilogue -- WITHOUT a launch: what mbx_conv would return short of a launch error.  For callers that keep a table of
 * measured tile choices and must not find out in the middle of a step that an entry no longer applies.               */
int mbx_conv_supported(const mbx_conv_desc* desc /*HOST*/);
/* TWO independent convolutions in ONE launch (sibling branches that do not fill the chip on their own: block35's two 3x3
 * convolutions, forward and data gradient).  Applies when both descriptors resolve to the same 4-wave 128x64 / 64x64 tile
 * with a store or store + statistics epilogue and general (non-pointwise, non stride-2-gradient) addressing; otherwise
 * MBX_ERR_UNSUPPORTED and nothing is launched (the caller then issues two mbx_conv).  Bit-identical to two mbx_conv.     */
int mbx_conv_pair(const mbx_conv_desc* a /*HOST*/, const mbx_conv_desc* b /*HOST*/, mbx_stream_t stream);
size_t mbx_conv_splitk_workspace_bytes(const mbx_conv_desc* desc /*HOST*/);   /* 0 unless tile_config is a split-K one */

/* Weight gradient (TF autodiff of slim.conv2d, train.py:263):
 * dw[k][r][s][c] += sum_{n,oh,ow} dy[n,oh,ow,k] * x[n, oh*stride-pad_t+r, ow*stride-pad_l+s, c]
 * dw is float32 KRSC, ACCUMULATED with atomics (zero it first).  desc gives x / geometry as
 * in the forward call; dy is an [N,H_out,W_out,C_out] bf16 view.  `db` (float32 [C_out],
 * may be NULL) accumulates sum dy for a bias gradient.                                     */
int mbx_conv_wgrad(const mbx_conv_desc* desc /*HOST: x, geometry, C_out*/, const void* dy,
                   int64_t dy_img_stride, int32_t ld_dy, float* dw, float* db, mbx_stream_t stream);

/* GROUPED weight gradient: the weight gradients of many layers (a whole backward segment) in ONE launch.  dW is
 * only consumed by the optimiser, so the caller keeps each layer's dy alive and defers the weight gradients to the
 * end of a segment: the launch walks a table of work items (layer, output tile, pixel range) built once by
 * mbx_wgrad_plan.  With hundreds of tiles in flight a tile's pixel reduction is split across blocks only when it
 * is longer than a fair share of one CU's work, so most dw elements have a single adder; flag
 * MBX_WGRAD_DETERMINISTIC forbids every split (bit-reproducible dw).  Semantics per job = mbx_conv_wgrad_scaled,
 * with one difference: dw / db MUST be zero before the launch and each job needs its own dw -- a tile whose pixel
 * reduction is not split is written with plain stores (one adder: nothing to add to), split tiles add atomically.
 * The launch is PERSISTENT: one 1024-thread workgroup per CU (8 MFMA waves + 8 LDS-DMA loader waves, 128 VGPRs per lane,
 * 144 KB + 16 B of dynamic LDS: three ring stages of up to six 8 KB sub-images; nothing else fits beside it) pulls work items
 * from per-XCD queues inside the image (the queue heads start at zero in the image and the kernel's last block
 * resets them, so the same image serves every launch; one launch of an image at a time).
 * mbx_wgrad_plan writes a HOST image of mbx_wgrad_plan_bytes() bytes; copy it to 16-byte aligned device memory
 * once and pass that pointer to every launch.  The image embeds the jobs' device pointers.                    */
typedef struct {
  mbx_conv_desc desc;      /* x, geometry, C_out as for mbx_conv_wgrad */
  const void* dy; int64_t dy_img_stride; int32_t ld_dy;
  float scale;             /* multiplies this job's dw / db contributions */
  float* dw; float* db;    /* float32, ACCUMULATED (zero first); db may be NULL */
} mbx_wgrad_job;
typedef struct {
  int32_t n_layers, n_items;
  int64_t layers_off, items_off;   /* byte offsets of the two tables inside the image */
  int64_t queues_off, heads_off;   /* per-XCD work queues: item ranges, and the queue heads the launch advances */
  double flops;                    /* 2 * M * C_out * R*S*C_in summed over the jobs */
  int64_t tally_off;               /* two uint64 the launches only ever add to: work items processed, launches completed;
                                      after a synchronise items == launches * n_items, or a launch has skipped work */
} mbx_wgrad_plan_info;
#define MBX_WGRAD_DETERMINISTIC 1
#define MBX_WGRAD_SCATTER 2        /* A/B knob: deal single items round-robin to the queues (no L2 panel sharing) */
size_t mbx_wgrad_plan_bytes(const mbx_wgrad_job* jobs /*HOST*/, int n_jobs, int flags);
int mbx_wgrad_plan(const mbx_wgrad_job* jobs /*HOST*/, int n_jobs, int flags, void* host_image, size_t bytes,
                   mbx_wgrad_plan_info* info /*HOST, out*/);
int mbx_conv_wgrad_grouped(void* device_image /*queue heads are written*/, const mbx_wgrad_plan_info* info /*HOST*/,
                           mbx_stream_t stream);
/* The same launch with at most max_workgroups persistent workgroups (<= 0: one per CU).  The weight gradient is off the
 * critical path of the backward pass (only the optimiser reads dW): launched on a second stream with a capped grid it
 * runs BESIDE the next segment's data-gradient / batch-norm chain on the CUs that chain leaves idle; any number of
 * workgroups drains the queues (placement is for speed only), results are those of the uncapped launch.            */
int mbx_conv_wgrad_grouped_capped(void* device_image, const mbx_wgrad_plan_info* info /*HOST*/, int max_workgroups,
                                  mbx_stream_t stream);

/* Scalars on the dgrad/wgrad path: mbx_conv multiplies the accumulator by `rscale` when
 * epilogue == MBX_EPI_STORE and rscale != 0 (the residual branch scale of model.py:21 on the
 * way back); mbx_conv_wgrad_scaled multiplies dw/db contributions by `scale`.             */
int mbx_conv_wgrad_scaled(const mbx_conv_desc* desc, const void* dy, int64_t dy_img_stride,
                          int32_t ld_dy, float scale, float* dw, float* db, mbx_stream_t stream);

/* ----------------------------------------------------------------- batch norm (K8, A3)
 * slim.batch_norm as the reference configures it (train.py:94-99): no gamma
 * (scale=False), beta, epsilon 0.001, batch statistics over N*H*W when training, moving
 * averages updated as moving -= (1-decay)*(moving - batch) (biased variance).
 * Forward training = mbx_conv(stats_partial) -> mbx_bn_finalize -> mbx_bn_apply.          */
/* decay < 0 ("store mode", also mbx_bn_apply_fused): moving_mean / moving_var are OVERWRITTEN with the batch mean and the
 * biased batch variance instead of being updated -- for callers that apply the moving-average update later, gated by
 * the step control word (mbx_bn_moving_update below), so that a step the optimiser skips leaves them untouched.      */
int mbx_bn_finalize(const float* stats_partial /*[rows,C,2]*/, int rows, int C, int64_t count,
                    float eps, float decay, float* mean /*[C]*/, float* rstd /*[C]*/,
                    float* moving_mean /*[C] or NULL*/, float* moving_var /*[C] or NULL*/,
                    mbx_stream_t stream);
/* a = relu?((y - mean) * rstd + beta): y bf16 [M,C] contiguous -> a bf16 view (ld_a).     */
int mbx_bn_apply(const void* y, int64_t M, int C, const float* mean, const float* rstd,
                 const float* beta, int relu, void* a, int ld_a, mbx_stream_t stream);
/* mbx_bn_finalize + mbx_bn_apply in ONE launch (each workgroup re-reduces the partials of its own 64
 * channels; used when rows <= 16, otherwise it issues the two launches: re-reducing hundreds of partial
 * rows per workgroup measured slower than the extra launch).  Same results either way.                          */
int mbx_bn_apply_fused(const float* stats_partial, int rows, int64_t count, float eps, float decay,
                       const void* y, int64_t M, int C, const float* beta, int relu, void* a, int ld_a,
                       float* mean, float* rstd, float* moving_mean, float* moving_var,
                       mbx_stream_t stream);
/* Finalize + apply in one launch for a convolution that ADDED its statistics into few rows (mbx_conv_desc.stats_rows_mod:
 * stats_fixed = that INT64 fixed-point table [rows][C][2], rows <= 16, C <= 2048; every workgroup reduces the rows of all
 * C channels itself, then sweeps whole rows of y), with the activation view addressed through a group's channel map (NULL:
 * the identity; see BATCH-NORM GROUPS below).
 * relu_thr ([C], may be NULL): the ReLU threshold on y, mean - beta / rstd (-inf without relu): (y - mean) rstd + beta > 0
 * <=> y > relu_thr -- what mbx_conv's BN-backward statistics epilogue masks with.                                      */
int mbx_bn_apply_fused_mapped(const void* stats_fixed, int rows, int64_t count, float eps, float decay,
                              const void* y, int64_t M, int C, const float* beta, int relu, void* a, int ld_a,
                              const struct mbx_chan_map_s* a_map /*HOST*/, float* mean, float* rstd, float* moving_mean,
                              float* moving_var, float* relu_thr, mbx_stream_t stream);
/* moving -= (1-decay)*(moving - batch) for n channels (every batch-norm layer of a step in one launch; batch_mean /
 * batch_var as written by the store mode above: the same float32 expressions, bit-identical to the in-place update).
 * skip_ctl (DEVICE, two float32, may be NULL): the step control block of mbx_rmsprop_ema_step -- if either word is
 * non-zero the launch changes nothing except *skipped_steps += 1 (DEVICE uint64, may be NULL): a poisoned step or a
 * step after a stop request is skipped EVERYWHERE, moving statistics included (train.py:94-99 updates them as part of
 * the train op, which the reference's py_func error aborts as a whole, loss.py:82).
 * ema_mean / ema_var (may be NULL): the ExponentialMovingAverage shadows of the two (train.py:253-259), updated from the
 * NEW values in the same launch: ema -= (1-ema_decay)*(ema - moving) -- mbx_ema_update's expression.                  */
int mbx_bn_moving_update(float* moving_mean, float* moving_var, const float* batch_mean, const float* batch_var,
                         int64_t n, float decay, const float* skip_ctl, uint64_t* skipped_steps,
                         float* ema_mean, float* ema_var, float ema_decay, mbx_stream_t stream);
/* BATCH-NORM GROUPS.  Sibling convolutions (same pixels, independent inputs: the two 3x3 branches of a block35,
 * model.py:11-17; the stride-2 branches of Mixed_7a, model.py:166-178) are normalised by ONE set of launches: their
 * pre-BN outputs are channel slices of one contiguous [M, C] tensor (each convolution writes its slice: ldy = C), beta /
 * mean / rstd / moving statistics are contiguous in member order, and the activation / gradient VIEW places the members'
 * channels wherever the concat layout wants them: channel c of the [M, C] tensor is channel c + offset[i] of the view,
 * i = the last entry with c_begin[i] <= c.  c_begin ascending from 0, everything a multiple of 8, offsets >= 0 (the
 * view pointer is that of the lowest member).  map = NULL: the identity.  A kernel costs >= 4.4 us in the replayed
 * step whatever it does; a group of two saves a finalize, an apply and a backward launch per step and block.          */
typedef struct mbx_chan_map_s { int32_t n; int32_t c_begin[4]; int32_t offset[4]; } mbx_chan_map;
/* mbx_bn_finalize over up to 4 members: member p's convolution wrote parts[p] = [rows[p]][Cs[p]][2]. */
int mbx_bn_finalize_parts(const float* const* parts /*HOST array of DEVICE pointers*/, const int32_t* rows /*HOST*/,
                          const int32_t* Cs /*HOST*/, int n_parts, int64_t count, float eps, float decay, float* mean,
                          float* rstd, float* moving_mean, float* moving_var, mbx_stream_t stream);
int mbx_bn_apply_mapped(const void* y, int64_t M, int C, const float* mean, const float* rstd, const float* beta,
                        int relu, void* a, int ld_a, const mbx_chan_map* a_map /*HOST*/, mbx_stream_t stream);
/* mbx_bn_apply followed by mbx_maxpool_fwd (3x3, stride 2, VALID) in ONE pass, for a layer whose activation feeds only that
 * pool (the two stem pools, model.py:103,115): p[n,oh,ow,c] = max over the window of bf16(relu?((y - mean) rstd + beta)),
 * argmax = the first maximum's tap (uint8 [N,Ho,Wo,C], may be NULL) -- the activation itself is never stored.
 * y: bf16 [N*H*W, C] contiguous.  Bit-identical to the two calls.                                                       */
int mbx_bn_apply_maxpool(const void* y, int N, int H, int W, int C, const float* mean, const float* rstd, const float* beta,
                         int relu, void* p, int64_t p_img_stride, int ld_p, int Ho, int Wo, uint8_t* argmax,
                         mbx_stream_t stream);
/* Frozen BN folded into the conv epilogue (detect.py:313-326, train.py:124-131):
 * scale = 1/sqrt(moving_var+eps), shift = beta - moving_mean*scale.                       */
int mbx_bn_fold(const float* moving_mean, const float* moving_var, const float* beta, float eps,
                int C, float* scale, float* shift, mbx_stream_t stream);
/* Backward through relu + batch norm.  xhat = (y-mean)*rstd;  g = da * (a > 0) (relu) -- or, when
 * relu != 0 and a == NULL, g = da * (xhat + beta > 0): the mask is recomputed from y with the forward
 * expression, so the activation is not read at all (`beta` is only needed for that form).
 * pass 1 writes partial sums {sum g, sum g*xhat} [rows,C,2]; mbx_bn_bwd_finalize reduces them,
 * ACCUMULATES dbeta += sum g and stores m1 = sum g / M, m2 = sum g*xhat / M;
 * pass 2: dy = rstd * (g - m1 - xhat*m2)  (bf16 [M,C]).                                          */
int mbx_bn_bwd_rows(int64_t M, int C);
int mbx_bn_bwd_reduce(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y,
                      int64_t M, int C, const float* mean, const float* rstd, const float* beta,
                      float* partial /*[mbx_bn_bwd_rows,C,2]*/, mbx_stream_t stream);
int mbx_bn_bwd_finalize(const float* partial, int rows, int C, int64_t M, float* dbeta /*[C] +=*/,
                        float* m12 /*[2,C]*/, mbx_stream_t stream);
int mbx_bn_bwd_apply(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y,
                     int64_t M, int C, const float* mean, const float* rstd, const float* beta,
                     const float* m12, void* dy /*bf16 [M,C]*/, mbx_stream_t stream);

/* ... with the gradient view `da` addressed through a group's channel map (a must be NULL: mask from y) */
int mbx_bn_bwd_reduce_mapped(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y, int64_t M, int C,
                             const float* mean, const float* rstd, const float* beta, float* partial,
                             const mbx_chan_map* da_map /*HOST*/, mbx_stream_t stream);
int mbx_bn_bwd_apply_mapped(const void* da, int ld_da, const void* a, int ld_a, int relu, const void* y, int64_t M, int C,
                            const float* mean, const float* rstd, const float* beta, const float* m12, void* dy,
                            const mbx_chan_map* da_map /*HOST*/, mbx_stream_t stream);

/* Backward through relu + batch norm as ONE STREAMING launch, for a layer whose sums {sum g, sum g y} were ADDED into
 * `rows` rows of stats = [rows][C][2] by the data gradient(s) that wrote da (mbx_bn_bwd_stats above): every workgroup
 * reduces the rows of all C channels itself, then dy = rstd (g - m1 - xhat m2) with g = (y > relu_thr) ? da : 0,
 * m1 = sum g / M, m2 = rstd (sum g y - mean sum g) / M; dbeta [C] += sum g (may be NULL).  No grid barrier, no
 * workspace to time out on: the same launch with and without a concurrent stream.  da through a group's channel map
 * (NULL: identity).  C <= 2048, rows <= 16.                                                                       */
int mbx_bn_bwd_apply_rows(const float* stats, int rows, const void* da, int ld_da, const void* y, int64_t M, int C,
                          const float* mean, const float* rstd, const float* relu_thr, float* dbeta, void* dy,
                          const mbx_chan_map* da_map /*HOST*/, mbx_stream_t stream);

/* The same backward pass in ONE launch (relu mask recomputed from y): every workgroup keeps its slice of
 * (da, y) in registers across a grid barrier, so da and y are read once.  Available when the layer fits
 * one resident workgroup per CU (mbx_bn_bwd_onepass_supported; everything but the 5 stem layers at
 * BATCH_SIZE 64); otherwise MBX_ERR_UNSUPPORTED and the caller uses the three launches above.
 * `ws` (mbx_bn_bwd_onepass_workspace_bytes(C), 16-byte aligned) must be ZERO at launch; after the launch word
 * [8*2*C] (behind the eight accumulator copies) holds the grid size and word [8*2*C + 1] a barrier-timeout flag (0 unless the grid was not resident);
 * a workgroup that timed out also writes NaN into its part of dy (and dbeta), so a step cannot continue silently on
 * partial totals.  dbeta [C] += sum g (may be NULL).  max_workgroups: 0 = one
 * workgroup per CU; a smaller positive number leaves CUs free for a concurrent stream (e.g. an RCCL
 * all-reduce in flight), whose kernels would otherwise delay the barrier until they finish.
 * step_poison (float32 scalar, may be NULL): += 1 per workgroup that timed out -- the first word of the step control
 * block that mbx_rmsprop_ema_step / mbx_ema_update test (skip_ctl), so that a poisoned step is never applied.      */
size_t mbx_bn_bwd_onepass_workspace_bytes(int C);
int mbx_bn_bwd_onepass_supported(int64_t M, int C, int max_workgroups);
int mbx_bn_bwd_onepass(const void* da, int ld_da, int relu, const void* y, int64_t M, int C,
                       const float* mean, const float* rstd, const float* beta, float* dbeta /*[C] +=*/,
                       void* dy /*bf16 [M,C]*/, void* ws, int max_workgroups, float* step_poison /*or NULL*/,
                       mbx_stream_t stream);

/* Three-launch backward of a layer whose activation feeds ONLY a 3x3 / stride-2 VALID max-pool (the two stem pools,
 * model.py:103,115: 177 MB and 124 MB tensors): the activation gradient is gathered from the pool's output gradient gy
 * [N,Ho,Wo,C] and its argmax bytes on the fly (mbx_maxpool_bwd's arithmetic, rounded to bf16 as it would store it: the
 * per-element values are those of mbx_maxpool_bwd; the statistics partials are grouped by 2 x 2 pixel blocks, so the
 * sums agree with mbx_bn_bwd_reduce to float32 rounding), so the max-pool backward launch, the write of its result and
 * the two reads of it disappear.  y / dy: [N*H*W, C] contiguous; relu mask from y; partial rows: mbx_bn_bwd_rows_pooled. */
int mbx_bn_bwd_rows_pooled(int N, int H, int W, int C);
int mbx_bn_bwd_reduce_pooled(const void* gy, int64_t gy_img_stride, int ld_gy, const uint8_t* argmax, int N, int H, int W,
                             int Ho, int Wo, int relu, const void* y, int C, const float* mean, const float* rstd,
                             const float* beta, float* partial /*[mbx_bn_bwd_rows_pooled(N,H,W,C), C, 2]*/, mbx_stream_t stream);
int mbx_bn_bwd_apply_pooled(const void* gy, int64_t gy_img_stride, int ld_gy, const uint8_t* argmax, int N, int H, int W,
                            int Ho, int Wo, int relu, const void* y, int C, const float* mean, const float* rstd,
                            const float* beta, const float* m12, void* dy, mbx_stream_t stream);
int mbx_bn_bwd_onepass_mapped(const void* da, int ld_da, int relu, const void* y, int64_t M, int C, const float* mean,
                              const float* rstd, const float* beta, float* dbeta, void* dy, void* ws, int max_workgroups,
                              float* step_poison, const mbx_chan_map* da_map /*HOST*/, mbx_stream_t stream);

/* ---------------------------------------------------------------------- pooling (K9)
 * NHWC bf16 views.  max: k x k, stride, VALID (model.py:103,115,157,180); argmax (uint8 tap
 * index, first maximum) is kept for the backward pass.  avg: k x k stride 1, pad `pad`
 * on every side, divisor = number of valid taps (TF SAME semantics, model.py:134; VALID
 * 8x8 of model.py:285 with pad 0).  Backward passes gather, `accumulate` adds to dx.      */
int mbx_maxpool_fwd(const void* x, int64_t x_img_stride, int ldx, int N, int H, int W, int C,
                    int k, int stride, void* y, int64_t y_img_stride, int ldy, int Ho, int Wo,
                    uint8_t* argmax /*[N,Ho,Wo,C] or NULL*/, mbx_stream_t stream);
int mbx_maxpool_bwd(const void* dy, int64_t dy_img_stride, int ld_dy, const uint8_t* argmax,
                    int N, int H, int W, int C, int k, int stride, int Ho, int Wo, void* dx,
                    int64_t dx_img_stride, int ld_dx, int accumulate, mbx_stream_t stream);
int mbx_avgpool_fwd(const void* x, int64_t x_img_stride, int ldx, int N, int H, int W, int C,
                    int k, int pad, void* y, int64_t y_img_stride, int ldy, int Ho, int Wo,
                    mbx_stream_t stream);
int mbx_avgpool_bwd(const void* dy, int64_t dy_img_stride, int ld_dy, int N, int H, int W, int C,
                    int k, int pad, int Ho, int Wo, void* dx, int64_t dx_img_stride, int ld_dx,
                    int accumulate, mbx_stream_t stream);

/* ------------------------------------------------------------------- small glue kernels
 * relu backward in place: g *= (a > 0)   (model.py:22-23 on the way back).                 */
int mbx_relu_mask(void* g, int ld_g, const void* a, int ld_a, int64_t M, int C, mbx_stream_t stream);
/* images float32 [N,H,W,3] -> bf16 [N,H,W,8] (channels 3..7 zero) for the stem conv.       */
int mbx_pack_input(const float* img, int64_t pixels, void* out, mbx_stream_t stream);
/* Detection heads (model.py:295-322): one head's float32 conv output h [N*cells, ld_h]
 * (channels [0,4k) locations, [4k,5k) confidence logits) <-> locations [N,P,4],
 * logits [N,P] at prior offset `off` (flatten order of model.py:296-319).
 * scatter writes the bf16 gradient [N*cells, ld_g] (zero beyond 5k) from d_locs/d_logits.  */
int mbx_head_gather(const float* h, int ld_h, int N, int cells, int k, int P, int off,
                    float* locs, float* logits, mbx_stream_t stream);
int mbx_head_scatter(const float* d_locs, const float* d_logits, int N, int cells, int k, int P,
                     int off, void* g, int ld_g, mbx_stream_t stream);

/* All heads of the network in ONE launch each way (at most 8; model.py:198-324 has six): the per-head calls above cost a
 * kernel launch each for a few kilobytes.  `h` / `ld_h` are read by the gather, `g` / `ld_g` written by the scatter.  */
typedef struct {
  const float* h; int32_t ld_h;   /* float32 conv output [N*cells, ld_h] */
  void* g; int32_t ld_g;          /* bf16 gradient [N*cells, ld_g] */
  int32_t cells, k, off;          /* grid cells, boxes per cell, prior offset of the head */
} mbx_head;
int mbx_head_gather_all(const mbx_head* heads /*HOST*/, int n_heads, int N, int P, float* locs, float* logits,
                        mbx_stream_t stream);
int mbx_head_scatter_all(const float* d_locs, const float* d_logits, const mbx_head* heads /*HOST*/, int n_heads, int N,
                         int P, mbx_stream_t stream);

/* Start of a training step's backward pass in one launch: grads[0, n_grads) and bn_ws[0, n_ws) (float32 counts, multiples
 * of 4, 16-byte aligned buffers) are cleared; before grads[ctl_index] -- word [0] of the step control block, the grid-
 * barrier time-outs of the PREVIOUS step (mbx_bn_bwd_onepass step_poison, summed over ranks) -- is cleared, its value is
 * added to *timeouts_total (DEVICE uint64, may be NULL), so that a time-out between two host checks is not lost.
 * zero_scalar (float32, may be NULL) is cleared too: the reg_loss accumulator mbx_rmsprop_ema_step adds to.              */
int mbx_step_begin(float* grads, int64_t n_grads, float* bn_ws, int64_t n_ws, int64_t ctl_index, uint64_t* timeouts_total,
                   float* zero_scalar, mbx_stream_t stream);

/* ---------------------------------------------------------- parameters (A8, K16-K18)
 * Filters live as float32 masters in one flat buffer (KRSC each); the bf16 copies the
 * kernels read are refreshed once per step.  mbx_filter_prepare writes, for every table
 * entry, the flipped channel-transposed dgrad copy [C][R][S][Kpad] from the bf16 KRSC copy. */
typedef struct {
  int64_t src_off;  /* element offset of the KRSC filter in the bf16 flat buffer           */
  int64_t dst_off;  /* element offset of the [C][R][S][Kpad] copy in the dgrad buffer       */
  int32_t K, R, S, C, Kpad;
  int32_t first_block; /* prefix sum of R*S*ceil(C/32)*ceil(Kpad/64) over the entries            */
} mbx_filter_entry;
int mbx_filter_prepare(const void* w_bf16, void* w_dgrad, const mbx_filter_entry* table /*DEVICE*/,
                       int n_entries, int total_blocks, mbx_stream_t stream);

/* RMSProp (tf.train.RMSPropOptimizer, train.py:207-212) + L2 regulariser gradient
 * (train.py:104-105: g += wd*w) + EMA of the variables (train.py:253-259) + refresh of the
 * bf16 copy, one pass over a flat parameter range:
 *   ema -= (1-ema_decay)*(ema - w)   [value before this step's update]
 *   g' = g + wd*w ; ms = decay*ms + (1-decay)*g'^2 ; mom = momentum*mom + lr*g'/sqrt(ms+eps)
 *   w -= mom ; w_bf16 = bf16(w)
 * reg_loss (float32 scalar, may be NULL) += wd/2 * sum w^2 (the value before the update:
 * slim's regularization loss in total_loss, train.py:246).  `trainable` = 0 skips the
 * update (frozen variables still get EMA, regulariser and bf16 refresh).
 * skip_ctl (DEVICE, two float32, may be NULL): the step control block -- [0] barrier timeouts of this step's
 * one-launch BN backward (mbx_bn_bwd_onepass step_poison), [1] ranks that asked to stop; data-parallel callers sum
 * it over ranks together with the gradients.  If either word is non-zero the launch does NOTHING (no update, no
 * EMA, no regulariser, no bf16 refresh): the step is skipped identically on every rank.      */
int mbx_rmsprop_ema_step(float* w, const float* g, float* ms, float* mom /*NULL if momentum==0*/,
                         float* ema /*or NULL*/, void* w_bf16 /*or NULL*/, int64_t n, float lr,
                         float decay, float momentum, float eps, float wd, float ema_decay,
                         int trainable, float* reg_loss, const float* skip_ctl, mbx_stream_t stream);
int mbx_ema_update(float* ema, const float* value, int64_t n, float ema_decay, const float* skip_ctl /*as above*/,
                   mbx_stream_t stream);

/* ------------------------------------------------------------ training input augmentation (row F1)
 * The pixel half of the reference's training input graph (inputs.py:264-351: crop, tf.image.resize_images with a
 * random ResizeMethod, distort_color, random_flip_left_right, (image - 0.5) * 2) for one batch.  The host draws every
 * random decision and decodes the JPEGs (multibox_amd/inputs.py: plan_augmentation); each item describes one image:
 * its already-cropped pixels [src_h][src_w][3] uint8 at src + src_offset, the resize method (0 bilinear, 1 nearest,
 * 2 bicubic, 3 area: TF 0.11's legacy kernels, align_corners = False; 4 = the bytes at src_offset are a float32
 * [S][S][3] picture in [0,1] the host prepared itself, src_offset a multiple of 4), the colour ops in application
 * order (0 brightness delta, 1 saturation factor, 2 hue delta, 3 contrast factor; n_ops = 0: no colour distortion and
 * no clipping), and the flip.  out[b] = the [S][S][3] float32 picture in [-1,1].
 * any_contrast must be non-zero iff some item has a contrast op (it adds the per-image mean pass).
 * Resize arithmetic is float32 in the host restatement's rounding order (bit-identical to it); colour ops run in
 * float64 like the restatement and differ from it only through the summation order of the contrast mean.          */
typedef struct {
  uint64_t src_offset;
  int32_t src_h, src_w;
  int32_t method;
  int32_t flip;
  int32_t n_ops;
  int32_t op[4];
  int32_t pad_;
  double arg[4];
} mbx_augment_item;                                   /* 80 bytes */
size_t mbx_augment_workspace_bytes(int B, int S);
int mbx_augment_batch(const uint8_t* src /*device*/, const mbx_augment_item* items /*device, [B]*/, int B, int S,
                      int any_contrast, float* out /*[B,S,S,3]*/, void* workspace, mbx_stream_t stream);

/* Detection input (row F3; detect.py:181-281): patch i is a win_h x win_w window at (win_y, win_x) of the decoded
 * image [img_h][img_w][3] uint8 at src + src_offset -- of its left-right mirror image when flip_source -- scaled to
 * [-1,1] first ((x/255 - 0.5) * 2, detect.py:181-182) and then resized to S x S with TF 0.11's legacy bilinear kernel
 * (tf.image.resize_images, align_corners = False).  The whole image as one window gives the original / flipped
 * original patches, the sliding windows of detect.extract_patches (detect.py:20-72) the crops.  The window must lie
 * inside the image (the host checks).  out[i] = float32 [S][S][3]; bit-identical to the host arithmetic.          */
typedef struct {
  uint64_t src_offset;
  int32_t img_h, img_w;
  int32_t win_y, win_x, win_h, win_w;
  int32_t flip_source;
  int32_t pad_;
} mbx_patch_item;                                     /* 40 bytes */
int mbx_extract_patches(const uint8_t* src /*device*/, const mbx_patch_item* items /*device, [n]*/, int n, int S,
                        float* out /*[n,S,S,3]*/, mbx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MBX_H */
