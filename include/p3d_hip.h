/* libp3dhip -- C ABI of the MI355X-native P3D saliency forward/backward path.
 *
 * The reference (A-Nasiri-M/sap3d_tensorflow) has no FFI layer: its boundary is the Python
 * graph function  p3d.p3d_unet(_X, _dropout, batch_size, training)  (reference p3d.py:169) plus
 * the tf.Session feed/fetch contract its drivers use (train.py:143-146,217-218,225-226;
 * gen_pred.py:45-46,151).  Each entry point below names the reference interface it replaces.
 * The binding a maintainer adds on the reference side is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * negative code on failure with the message available from p3d_last_error(); host buffers are
 * owned by the caller and copied by value (like feed_dict / fetched numpy arrays); device
 * memory, streams and RCCL state are owned by the opaque handle; one caller thread per handle
 * (like the single thread calling sess.run).  All tensors are float32 NDHWC.
 */
#ifndef P3D_HIP_H
#define P3D_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct p3d_handle p3d_handle;

enum {
    P3D_STRUCTURE_UNET = 0,      /* train.py:149-150  --structure unet   -> p3d.p3d_unet   (p3d.py:169) */
    P3D_STRUCTURE_CONCAT = 1,    /* train.py:151-152  --structure concat -> p3d.p3d_concat (p3d.py:224), no sigmoid */
    P3D_STRUCTURE_GN_P3D = 2,    /* gn/train_p3d_gn_dataset.py:169-170 net='P3D' -> p3d_gn.inference_p3d (gn/p3d_gn.py:214):
                                    GroupNorm + CBAM on every residual, concat head, no sigmoid */
    P3D_STRUCTURE_UNETPP_NONSA = 3, /* p3d.p3d_unetplusplus_nonsa (p3d.py:401): the nested UNet++ head of
                                    train.py:153-154 `--structure unet++` with its attention blocks left out
                                    (layer wrappers utils/network.py:97-110) */
    P3D_STRUCTURE_GN_P3D_DECODER = 4, /* gn/train_p3d_gn_dataset.py:177-178 net='P3D_DECODER' ->
                                    p3d_gn.inference_p3d_decoder_block (gn/p3d_gn.py:489): GN/CBAM encoder, skip
                                    deconvs + concat, two conv-deconv-conv decoder blocks, 3x3x3 conv to 1 channel;
                                    variables live in scope "P3D/"; base must be a multiple of 16 */
    P3D_STRUCTURE_GN_P3D_CONCAT = 5, /* gn/train_p3d_gn_dataset.py:171-172 net='P3D_CONCAT' -> p3d_gn.inference_p3d_concat
                                    (gn/p3d_gn.py:279): GN_P3D with deconv_pool4 at 8*base instead of 16*base filters */
    P3D_STRUCTURE_UNETPP_DS = 6   /* p3d.p3d_unetplusplus_ds (p3d.py:340): the UNet++ head WITH self attention
                                    (utils/network.py:157-192) on x_4_0, x_3_1, x_2_2 and, keys/values pooled by 2,
                                    x_1_3.  (train.py:153-154 wires p3d_unetplusplus, p3d.py:280, whose last
                                    attention call adds tensors of different shapes and cannot be built.)
                                    base must be a multiple of 16 */
};

typedef struct p3d_config {
    int structure;        /* P3D_STRUCTURE_*                                                    */
    int batch;            /* clips per step on THIS device (placeholder dim 0, train.py:143)    */
    int frames;           /* 16 in the reference (train.py:139, p3d.py:5); multiple of 16        */
    int height, width;    /* 112 in the reference (p3d.py:4); multiples of 16                   */
    int base;             /* stem width; 64 in the reference (p3d.py:172,179,185,191)           */
    int blocks[3];        /* bottlenecks per stage; 3,8,36 in the reference (p3d.py:179,185,191) */
    int device;           /* HIP device ordinal (train.py:73 CUDA_VISIBLE_DEVICES=args.gpu)      */
    int world_size;       /* data-parallel replicas (1 = the reference's single device)          */
    int rank;
} p3d_config;

/* Fill *cfg with the reference architecture: unet, batch 2, 16x112x112, base 64, blocks 3/8/36. */
void p3d_default_config(p3d_config* cfg);

/* Builds the graph once, like the graph-construction part of train.py:143-172 / gen_pred.py:45-46. */
int p3d_create(const p3d_config* cfg, p3d_handle** out);
void p3d_destroy(p3d_handle* h);
const char* p3d_last_error(void);

/* ---- variables: tf.global_variables(), Saver var_list (train.py:180-185).  Names are the TF
 *      variable names (firstconv1, conv3_0_1, STA_0_2_S, STA_0_2_S_bias, dw3d_0,
 *      batch_normalization_7/gamma, .../moving_mean, conv3d_transpose/kernel, deconv1_bn/beta ...). */
int p3d_num_params(p3d_handle* h);
int p3d_param_info(p3d_handle* h, int index, const char** name, int* ndim, int64_t shape[5], int* trainable);
int p3d_set_param(p3d_handle* h, const char* name, const float* host, int64_t count);   /* saver.restore */
int p3d_get_param(p3d_handle* h, const char* name, float* host, int64_t count);         /* saver.save    */
int p3d_get_grad(p3d_handle* h, const char* name, float* host, int64_t count);          /* tf.gradients (parity hook) */
/* tf.global_variables_initializer (train.py:178): Xavier-uniform etc., SURVEY.md Appendix A.7. */
int p3d_init_params(p3d_handle* h, uint64_t seed);

/* ---- sess.run(pred, {x, dropout: 0, training: False})   train.py:225-226, gen_pred.py:151.
 *      x [B,T,H,W,3] -> pred [B,T,H,W,1].  `training` is the placeholder of train.py:145: it
 *      switches stem/decoder BN and dropout only; backbone BN always uses batch statistics
 *      (p3d.py:140,179,185,191).  Never updates moving statistics (UPDATE_OPS are not fetched). */
int p3d_forward(p3d_handle* h, const float* x, int training, float dropout_rate, uint64_t seed, float* pred);

/* ---- B sliding windows of gen_pred.py:100-168 in one pass.  The reference runs sess.run(pred, ...) once per
 *      window with a batch of ONE clip (gen_pred.py:45,151), so the backbone's batch-statistics BatchNorm
 *      (p3d.py:140) sees one clip at a time.  This entry point takes `batch` windows x [B,T,H,W,3] and returns for
 *      each exactly what a batch-of-1 p3d_forward(training=0, dropout=0) returns for it: every batch-statistics
 *      BN normalises each clip with that clip's own statistics.  (For the GroupNorm structures it equals
 *      p3d_forward, GN has no cross-clip coupling.) */
int p3d_predict_windows(p3d_handle* h, const float* x, float* pred);

/* ---- sess.run([train_op, loss], {x, y, dropout, training: True})   train.py:217-218.
 *      y [B,T,H,W]; Smooth-L1 SUM loss (utils/network.py:49-62), Adam on every trainable
 *      (train.py:168), BN moving-average updates (train.py:170-172).  With world_size > 1 the
 *      gradients are summed across replicas (RCCL) before Adam. */
int p3d_train_step(p3d_handle* h, const float* x, const float* y, float dropout_rate, uint64_t seed, float* loss);

/* Parity hook: forward (training=True) + loss + backward, no Adam, no moving-stat update.
 * pred may be NULL.  Gradients are then readable with p3d_get_grad. */
int p3d_backward(p3d_handle* h, const float* x, const float* y, float dropout_rate, uint64_t seed,
                 float* loss, float* pred);

/* BASELINE.json configs[4] ("fp16 MFMA pointwise convs"; no reference counterpart, the reference is fp32 throughout):
 * when enabled, every 1x1x1 convolution (forward and input gradient) rounds its operands to fp16 in registers and
 * multiplies on the fp16 matrix cores with fp32 accumulation.  Everything stored stays fp32.  Parity for this mode is
 * fp16-level (2e-2 relative on the saliency maps); the default, fp32, is the mode the 1e-3 target applies to. */
int p3d_set_pointwise_fp16(p3d_handle* h, int enable);

/* BatchNorm fusion (no reference counterpart: an execution choice, the arithmetic is tf.layers.batch_normalization's either
 * way, p3d.py:56-81,88-97).  When on, the bn -> relu pairs between the convs of a small-tensor bottleneck are applied on the
 * operand paths of the neighbouring convolutions and never stored (fewer launches; measured no faster on one MI355X at 8
 * clips, so the default is off).  enable = 0: every BatchNorm is a pass of its own (the
 * round-2 launch list); 1: the forward pass is fused, BatchNorm's backward keeps its launches (the filter gradients read
 * the never-stored activations through the transform); 2: the backward pass is fused as well. */
int p3d_set_bn_fusion(p3d_handle* h, int enable);

/* The core of attention(), softmax(g f^T) h (utils/network.py:183-185; p3d_unetplusplus_ds, p3d.py:340-397) -- an execution
 * choice, the arithmetic is the reference's either way (only the order of the sums differs):
 *   1  three GEMMs per direction around a stored [N_g x N_f] score matrix and attention map (2 x B*N_g*N_f floats per block);
 *   2  score tiles recomputed on chip, nothing of size N_g x N_f in HBM (blocks of 32, 64, 128 or 256 channels; wider ones
 *      keep the GEMMs) -- the only way the last block fits at 8 clips of 32x224x224 (B*N_g*N_f = 5e10);
 *   0  (default) per block: 2 where the score matrix has 2^24 elements or more, else 1.
 * Switching to 1 allocates the score buffers of blocks that were built without them. */
int p3d_set_attention_mode(p3d_handle* h, int mode);

/* tf.train.AdamOptimizer(lr, beta1, beta2, epsilon) (train.py:168; defaults 1e-4, .9, .999, 1e-8). */
int p3d_set_adam(p3d_handle* h, float lr, float beta1, float beta2, float eps);

/* ---- intermediate tensors (tf fetches of graph tensors; parity/debug taps).  Names:
 *      conv1_custom, conv1_custom_bn_relu, pool1..pool4, block<i>/conv1_bn_relu, block<i>/st,
 *      block<i>/out, deconv3_re, deconv4_conv1, logits, pred. */
int p3d_activation_info(p3d_handle* h, const char* name, int64_t shape[5]);
int p3d_get_activation(p3d_handle* h, const char* name, float* host, int64_t count);

/* ---- schedule of one train step (TEST HOOK, tests/test_gpu_schedule.py): runs p3d_train_step_device once and writes every stream
 *      operation it issued, in host issue order, one per line: "L <stream> <kernel> [@op]" (launch), "M <stream> <what>" (async
 *      fill), "R <stream> e<k>" (event record), "W <stream> e<k>" (stream waits for event), "C <stream> allreduce <lo> <hi>".
 *      Streams: main, side (filter gradients), comm (all-reduce).  `needed` receives the size of the text; call with a buffer
 *      at least that large (a first call with text = NULL runs the step, too).  Synchronises. */
int p3d_debug_schedule(p3d_handle* h, float dropout_rate, uint64_t seed, char* text, int64_t cap, int64_t* needed);

/* ---- decisions of the last forward (TEST HOOK, tests/test_gpu_pinned.py): the ReLU gates and max-pool choices the backward pass of
 *      this handle will use -- so that the oracle can differentiate the SAME piecewise-linear branch (a float32 forward takes
 *      a handful of near-zero decisions differently from a float64 one, and each moves a gradient tensor by per cent).
 *      Site `index` (0 .. count-1, graph order) is a normalise / ReLU pass ("bn": out1 / out2 are NON-ZERO where the gate of the
 *      first / second BatchNorm branch -- TF scopes name1 / name2, name2 empty when there is one -- is open; produced by the
 *      pass's own backward kernel on a gradient of ones with the statistics terms off) or a max-pool ("pool": out1 is the pool's
 *      input as this handle holds it; its arg-max follows from it exactly).  count = elements of `shape`.  Call after a forward
 *      or backward pass of the whole graph with BatchNorm fusion off; synchronises. */
int p3d_debug_decision_count(p3d_handle* h);
int p3d_debug_decision_info(p3d_handle* h, int index, const char** kind, const char** name1, const char** name2, int64_t shape[5]);
int p3d_debug_decision_get(p3d_handle* h, int index, float* out1, float* out2, int64_t count);

/* ---- one bottleneck in isolation (BASELINE.json configs[0], SURVEY.md 8d cfg 1 "standalone variant"): the forward of
 *      Bottleneck(...).infer() (p3d.py:83-136; gn/p3d_gn.py:127-179 for the GN structures) number `block_id`
 *      (0 .. sum(blocks)-1) of this handle's graph on a caller-supplied input, with the handle's current parameters.
 *      Shapes are those the block has inside the graph (p3d_block_info).  BatchNorm uses batch statistics, as
 *      everywhere in the backbone. */
int p3d_block_info(p3d_handle* h, int block_id, int64_t in_shape[5], int64_t out_shape[5]);
int p3d_block_forward(p3d_handle* h, int block_id, const float* in, int64_t in_count, float* out, int64_t out_count);
/* The same bottleneck, forward then backward: `dout` is the gradient of its output, `din` receives the gradient of its input; the
 * gradients of the block's own variables are read with p3d_get_grad afterwards (every other variable's gradient reads 0).
 * Parity hook for gradients at sizes where the whole graph is out of the oracle's reach. */
int p3d_block_backward(p3d_handle* h, int block_id, const float* in, int64_t in_count, const float* dout, int64_t out_count, float* din);

/* ---- device-resident stepping for measurement: inputs already in HBM (bench.py).
 *      p3d_device_inputs returns the handle's own x / y staging buffers (device pointers,
 *      [B,T,H,W,3] and [B,T,H,W] floats); fill them once with p3d_upload_inputs, then call
 *      p3d_train_step_device / p3d_forward_device repeatedly; p3d_synchronize drains the stream. */
int p3d_upload_inputs(p3d_handle* h, const float* x, const float* y);
int p3d_train_step_device(p3d_handle* h, float dropout_rate, uint64_t seed);
int p3d_forward_device(p3d_handle* h, int training, float dropout_rate, uint64_t seed);
int p3d_last_loss(p3d_handle* h, float* loss);          /* synchronises */
int p3d_synchronize(p3d_handle* h);

/* ---- per-launch timing of one train step with HIP events on the stream the kernels are launched
 *      on (the data behind bench.py's roofline object).  One record per kernel launch, in launch
 *      order.  Writes up to `cap` records; returns the number of launches. */
typedef struct p3d_op_time {
    char name[64];        /* graph op the launch belongs to (block7/convS, deconv3, ...) */
    char kernel[48];      /* kernel symbol (igemm_kernel<128,128>, bn_apply_kernel<1>, ...) */
    double ms;            /* HIP-event duration of this launch */
    double flops;         /* algorithmic FLOPs of this launch (2*MACs) */
    double bytes;         /* algorithmic HBM bytes of this launch (operands read once, result written once) */
    int phase;            /* 0 forward, 1 backward, 2 optimiser */
} p3d_op_time;
int p3d_profile_step(p3d_handle* h, float dropout_rate, uint64_t seed, p3d_op_time* out, int cap);

/* ---- data parallel (no reference counterpart: the reference is single-device, train.py:73).
 *      Rank 0 creates an id, every rank passes the same bytes to p3d_comm_init. */
#define P3D_COMM_ID_BYTES 128
int p3d_comm_unique_id(void* id_out);
int p3d_comm_init(p3d_handle* h, const void* id);
/* GPUs this process can see (hipGetDeviceCount), -1 on error.  A launcher may give every rank ALL the node's GPUs (torch.distributed.run:
 * rank r uses device LOCAL_RANK) or exactly one (per-rank HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES: every rank uses device 0);
 * bench.py asks before it picks p3d_config.device. */
int p3d_device_count(void);
/* What RCCL itself says about the handle's communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): *n_ranks = 0 when the
 * handle has none.  bench.py prints it as "rccl_ranks" at N > 1 and refuses a run where it differs from --gpus, so that a scaling
 * line cannot come from ranks that never met. */
int p3d_comm_info(p3d_handle* h, int* n_ranks, int* rank, int* device);
/* Audit of the bucketed gradient hand-over (test hook; needs no communicator).  Runs forward + loss + backward with
 * buckets of `bucket_floats` and, at every point where a bucket [lo, hi) of the flat gradient buffer would be handed to
 * the all-reduce, waits for the work queued so far and copies the range out instead.  Returns the number of buckets
 * (their ranges and the op index after whose backward each was handed over go to lo / hi / after_op, up to `cap`),
 * *n_train = floats in the flat buffer, *stale = how many bucket elements differed from the final gradient, i.e.
 * were handed over before their last producer had run (must be 0). */
int p3d_debug_bucket_audit(p3d_handle* h, float dropout_rate, uint64_t seed, int64_t bucket_floats, int64_t* lo, int64_t* hi,
                           int32_t* after_op, int cap, int64_t* n_train, int64_t* stale);

/* Test hook: synchronises the device and returns how many arrival counters of the K-slice exchange scratch are non-zero
 * (every sliced launch re-zeroes its own: anything but 0 means a launch left the scratch dirty); -1 on a HIP error. */
int64_t p3d_debug_dirty_counters(void);

/* Test hook (process-wide): on SIGABRT / SIGSEGV print a C-level backtrace to stderr before dying, and the message of an uncaught
 * C++ exception (tests/conftest.py installs it: Python's faulthandler shows Python frames only). */
int p3d_debug_install_abort_trace(void);

/* Test hook (process-wide): force the tile / K-slice plan of the convolution kernels where a problem allows it, so that
 * every instantiation is reachable from the op-level parity tests.  igemm_tile: 0 = 64x64, 1 = 128x64, 2 = 128x128,
 * -1 = the plan's choice; igemm_splits: K-slices, 0 = the plan's; wgrad_tm / wgrad_tn: 64 or 128, 0 = the plan's. */
int p3d_debug_force_plan(int igemm_tile, int igemm_splits, int wgrad_tm, int wgrad_tn);

/* ---- single operators on host arrays (the TF ops the path is made of), for op-level parity
 *      tests.  SAME padding, NDHWC, filters [kd,kh,kw,Cin,Cout]; strides s[3] = (sd,sh,sw). */
int p3d_op_conv3d(int device, const float* x, const int64_t xshape[5], const float* w, const int64_t wshape[5],
                  const int s[3], const float* bias, float* y);                       /* tf.nn.conv3d (+bias_add) */
int p3d_op_conv3d_backprop_input(int device, const float* dy, const float* w, const int64_t wshape[5],
                                 const int s[3], const int64_t xshape[5], float* dx);
int p3d_op_conv3d_backprop_filter(int device, const float* x, const int64_t xshape[5], const float* dy,
                                  const int64_t wshape[5], const int s[3], float* dw, float* dbias);
/* tf.layers.conv3d_transpose 'same'; kernel [kd,kh,kw,Cout,Cin] */
int p3d_op_conv3d_transpose(int device, const float* x, const int64_t xshape[5], const float* k,
                            const int64_t kshape[5], const int s[3], const float* bias, float* y);
/* Test hook: the stem's filter gradient through its BatchNorm + ReLU (tf.gradients of p3d.py:172-174 w.r.t. firstconv1), once with
 * the normalisation's backward evaluated on the kernel's operand (what the train step runs) and once as two launches.  x [N,D,H,W,3];
 * y, dz [N,D,ceil(H/2),ceil(W/2),64]; tab [5][64] = scale, shift, mean, invstd, gamma; coef [64][2]; both results [1,7,7,3,64]. */
int p3d_debug_stem_wgrad_through_bn(int device, const float* x, const int64_t xshape[5], const float* y, const float* dz, const float* tab,
                                    const float* coef, int batch, float* dw_fused, float* dw_two_launches);
int p3d_op_max_pool3d(int device, const float* x, const int64_t xshape[5], const int ksize[3], const int s[3], float* y);
int p3d_op_max_pool3d_grad(int device, const float* x, const int64_t xshape[5], const int ksize[3], const int s[3],
                           const float* dy, float* dx);
/* BiasAddGrad of tf.nn.bias_add (the bias of tf.layers.conv3d / conv3d_transpose, p3d.py:147-150): dbias[c] = sum over
 * rows of dy[row][c]; fixed summation order, bit-identical run to run. */
int p3d_op_bias_add_grad(int device, const float* dy, int64_t rows, int channels, float* dbias);
/* The core of attention(), utils/network.py:183-185, on flattened operands:
 *     s = tf.matmul(hw_flatten(g), hw_flatten(f), transpose_b=True); beta = tf.nn.softmax(s); o = tf.matmul(beta, hw_flatten(h))
 * g [batch][n_g][ch/8], f [batch][n_f][ch/8], h [batch][n_f][ch] -> o [batch][n_g][ch], on the kernels that keep the score
 * matrix on chip (ch in {32, 64, 128, 256}).  With d_o (the gradient of o) it also writes dg, df, dh. */
int p3d_op_attention_core(int device, int batch, int n_g, int n_f, int ch, const float* g, const float* f, const float* h,
                          float* o, const float* d_o, float* dg, float* df, float* dh);

/* ---- validation metrics and frame pre-processing: the steps either side of the path.
 *      Host arrays in and out, float64 results.  Maps are float32 [n_maps][n_pix] of ONE shape (the reference's
 *      resize-to-match branch is not on the trainers' path).  Reference: utils/metrics.py. */
int p3d_metric_cc(int device, const float* map1, const float* map2, int n_maps, int n_pix, double* out);    /* CC, utils/metrics.py:227-250 */
int p3d_metric_sim(int device, const float* map1, const float* map2, int n_maps, int n_pix, double* out);   /* SIM, :258-287 */
int p3d_metric_nss(int device, const float* sal, const float* fix, int n_maps, int n_pix, double* out);     /* NSS, :200-224; fix > 0.5 */
/* AUC_Judd, utils/metrics.py:25-85.  jitter: the noise the reference adds (random.rand * 1e-7, :62-63) supplied by the
 * caller, [n_maps][n_pix], or NULL for jitter=False.  NaN where a map has no fixation. */
int p3d_metric_auc_judd(int device, const float* sal, const float* fix, const float* jitter, int n_maps, int n_pix, double* out);
/* AUC_Borji, utils/metrics.py:88-154, one map.  rand_idx = the reference's r = random.randint(0, n_pix, [n_fix, n_rep])
 * (:139), row-major; n_fix must equal the number of pixels with fix > 0.5.  out[n_rep] = area per random split (the
 * reference returns their mean). */
int p3d_metric_auc_borji(int device, const float* sal, const float* fix, const int* rand_idx, int n_pix, int n_fix, int n_rep,
                         double step_size, double* out);
/* mapf, dataflow.py:198-216: n decoded BGR uint8 frames [n][H0][W0][3] -> RGB - mean_rgb -> cv2.INTER_LINEAR resize to
 * H x W -> / 255, float32 [n][H][W][3] (one clip of the NDHWC input); and the grey density maps [n][H0][W0] -> [n][H][W]. */
int p3d_mapf_frames(int device, const unsigned char* bgr, int n, int H0, int W0, const float mean_rgb[3], int H, int W, float* out);
int p3d_mapf_density(int device, const unsigned char* grey, int n, int H0, int W0, int H, int W, float* out);

/* CRC-32C of a host buffer (host-side helper of the TensorFlow checkpoint reader / writer, sap3d_tensorflow_amd/tf_checkpoint.py:
 * the bundle format of train.py:180-185,266-267 checksums every tensor); crc = running value, 0 to start. */
uint32_t p3d_crc32c(const void* data, size_t n, uint32_t crc);

/* Releases every process-wide device resource of the library (scratch pools, the zero page) and synchronises the
 * device; live handles must be destroyed first.  The Python shim calls it from an atexit hook so that nothing of the
 * library is left for static destructors that may run after the HIP runtime is gone. */
int p3d_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
