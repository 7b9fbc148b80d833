/* cistaflow.h -- C ABI of libcistaflow.so, the MI355X (gfx950) hot path of CISTA-Flow.
 *
 * The reference (lsying009/CISTA-Flow) is pure Python/PyTorch and has no FFI of its own; the
 * boundary this library replaces is the Python module API of
 *     e2v/e2v_model.py:10-98    CistaLSTCNet.forward              -> cf_cista_forward
 *     DCEIFlow/DCEIFlow.py:143  DCEIFlow.forward                  -> cf_flow_forward
 *     utils/flow_utils.py:193   FrameWarp.warp_frame              -> cf_warp
 *     e2v/e2v_model.py:144-196  DCEIFlowCistaNet.forward (a5)     -> cf_step
 * The Python shells in cista_flow_amd/ bind these entry points with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data owned by the caller (PyTorch); the library
 *     owns only its per-handle workspace arena and the packed weight copies;
 *   - "NCHW" tensors are contiguous [B][C][H][W]; "NHWC" tensors are contiguous [B][H][W][C]
 *     (torch channels_last memory of a logical [B,C,H,W] tensor);
 *   - calls are asynchronous on `stream` (a hipStream_t), never synchronise the device, are not
 *     re-entrant per handle, and introduce no host round trip (the reference's per-frame
 *     `flow_final.any()` sync, e2v_model.py:184, becomes a device-side flag);
 *   - return value: 0 = ok, negative = error (message via cf_last_error); no exceptions cross.
 */
#ifndef CISTAFLOW_H
#define CISTAFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cf_handle cf_handle;

enum { CF_MODE_CISTA = 0, CF_MODE_EIFLOW = 1, CF_MODE_ERAFT = 2, CF_MODE_IDNET = 3 };
enum { CF_WARP_FORWARD = 0, CF_WARP_BACKWARD = 1 };

enum {
    CF_OK = 0,
    CF_ERR_ARG = -1,      /* bad argument / shape */
    CF_ERR_HIP = -2,      /* HIP runtime error */
    CF_ERR_STATE = -3,    /* call order (e.g. forward before weights were finalised) */
    CF_ERR_WEIGHT = -4,   /* missing / mis-shaped weight */
    CF_ERR_UNSUPPORTED = -5
};

typedef struct cf_config {
    int mode;           /* CF_MODE_* */
    int batch;          /* B: independent sequences held by this handle */
    int height, width;  /* image_dim (utils/configs.py:6); must be even */
    int num_bins;       /* 5  (configs.py:18); the 7x7 encoder stems read the voxel grid through the planar gather
                           convolution, whose K = 49*num_bins must be <= 256: num_bins <= 5 for eiflow / eraft
                           (cf_finalize_weights fails with CF_ERR_UNSUPPORTED beyond that), <= 28 for CISTA alone */
    int base_channels;  /* 64 (configs.py:22) */
    int depth;          /* 5 ISTA iterations (configs.py:20) */
    int iters;          /* flow-net refinement iterations: 6 eiflow (DCEIFlow.py:143), 12 eraft */
    int warp_mode;      /* CF_WARP_* (configs.py:94) */
    int device;         /* HIP device ordinal */
    int precision;      /* arithmetic of the convolution products (fp32 tensors and accumulation in every mode):
                           0 = exact fp32 (v_mfma_f32_32x32x2_f32)                              -- the default
                           3 = "f16x3": operands split hi+lo into f16, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16
                               (22-bit products, ~2e-5 end-to-end vs the reference)
                           1 = plain f16 products (reduced precision; BASELINE configs[4] "fp16 with MFMA") */
} cf_config;

/* lifetime ---------------------------------------------------------------------------------- */
int cf_create(cf_handle** out, const cf_config* cfg);
void cf_destroy(cf_handle* h);
const char* cf_last_error(const cf_handle* h);   /* h may be NULL: last create() error */
size_t cf_workspace_bytes(const cf_handle* h);

/* weights: announce every state_dict entry (name = reference state_dict key, fp32, contiguous,
 * PyTorch layout e.g. OIHW), then cf_finalize_weights packs them on `stream` (BatchNorm folded,
 * K-contiguous [cout][tap][cin] matrices).  The announced pointers are only read during
 * cf_finalize_weights.  Replaces nn.Module.load_state_dict + .to(device) for the hot path.
 * cf_finalize_weights is the ONE entry point that synchronises: it ends with hipStreamSynchronize(stream), because the announced
 * pointers may die as soon as it returns (the packing kernels must have read them).  It runs once per weight change, never per
 * frame; every forward entry point below is asynchronous on its stream. */
int cf_load_weights(cf_handle* h, const char* name, const void* dev_ptr, const int64_t* shape, int ndim);
int cf_finalize_weights(cf_handle* h, void* stream);

/* a4  FrameWarp.warp_frame (utils/flow_utils.py:212-221).  img/out NHWC [B][H][W][C] (C == 1:
 * identical to NCHW), flow NCHW [B][2][Hf][Wf]; (Hf,Wf) != (H,W) resamples the flow first with
 * interpolate(bilinear, align_corners=True) without rescaling values (e2v_model.py:190). */
int cf_warp(cf_handle* h, const float* img, const float* flow, float* out, int B, int C, int H, int W,
            int Hf, int Wf, int mode, void* stream);

/* a3  CistaLSTCNet.forward (e2v/e2v_model.py:49-98).
 *   ev NCHW [B][bins][H][W], img NCHW [B][1][H][W];
 *   states NHWC at (H/2,W/2): c,z 2*base channels; h,cc base channels; *_prev may be NULL (zeros).
 *   The *_out states must not alias the *_prev ones (the reference returns fresh tensors each frame too). */
int cf_cista_forward(cf_handle* h, const float* ev, const float* img, const float* c_prev, const float* z_prev,
                     const float* h_prev, const float* cc_prev, float* I_out, float* c_out, float* z_out,
                     float* h_out, float* cc_out, void* stream);

/* a13/a14  flow network forward.
 *   eiflow: in0 = event_voxel [B][bins][H][W], in1 = image1 [B][1][H][W]   (DCEIFlow.py:143)
 *   flow_init NCHW [B][2][Hp/8][Wp/8] or NULL; flow_final NCHW [B][2][H][W];
 *   flow_low (nullable) NCHW [B][2][Hp/8][Wp/8] = coords1 - coords0 ("flow_init" of the dict);
 *   flow_preds (nullable) NCHW [iters][B][2][Hp][Wp] = every iteration's padded up-sampled flow. */
int cf_flow_forward(cf_handle* h, const float* in0, const float* in1, const float* flow_init, float* flow_final,
                    float* flow_low, float* flow_preds, void* stream);

/* a14  ERAFT driver carry (test_with_flow.py:144-149: `event_voxel_old` of frame t is `event_voxel` of frame t-1).
 * Declares that in0 of the NEXT cf_step / cf_flow_forward holds the same bytes as in1 of the previous one, so fnet(in0)
 * is taken from the previous step instead of being recomputed (bit-identical: same kernels on the same input).
 * One-shot; a no-op for the other modes and when there is no previous step. */
int cf_hint_prev_grid(cf_handle* h, int same_as_previous_in1);

/* a5  one reconstructed frame: flow net -> any() -> warp I, warp Z -> CISTA-LSTC
 * (e2v/e2v_model.py:144-196).  gt_flow (nullable) overrides the estimated flow for the warp
 * (e2v_model.py:181-182).  z_warped_out (nullable unless z_prev != NULL) receives the warped
 * sparse code the reference stores back into the caller's states[1] (e2v_model.py:191). */
int cf_step(cf_handle* h, const float* in0, const float* in1, const float* rec_img0, const float* flow_init,
            const float* gt_flow, const float* c_prev, const float* z_prev, const float* h_prev,
            const float* cc_prev, float* I_out, float* flow_final, float* flow_low, float* flow_preds,
            float* z_warped_out, float* c_out, float* z_out, float* h_out, float* cc_out, void* stream);

/* hipGraph replay of cf_step (opt-in: cf_graph_enable(h, 1) or environment CF_GRAPH=1; measured no faster than the eager
 * launches on MI355X / ROCm 7.2 -- DESIGN.md section 8 -- so it is off by default).  A step is several hundred launches
 * over up to four streams; the library captures it once per distinct tuple of the 18 caller pointers (the second time
 * a tuple is seen) and replays the executable on a hit -- the same kernels with the same arguments, so results are
 * bit-identical to the eager path; callers whose buffers never repeat stay on the eager path.  Bypassed while
 * cf_profile_enable / CF_PHASES / CF_SERIAL are active and when `stream` is itself being captured.
 * cf_graph_stats: out3 = {captures, replays, executables cached}. */
int cf_graph_enable(cf_handle* h, int on);
int cf_graph_stats(const cf_handle* h, long long* out3);

/* f-1 (next row): events_to_voxel_grid + event_preprocess('std') (utils/event_process.py:15-72,193-216).
 * events: device [total][4] fp64 rows (timestamp, x, y, polarity), the B sequences' events concatenated in time
 * order; offsets: device int64 [B+1]; voxel: [B][bins][H][W] fp32; stats_scratch: 3*B doubles. */
int cf_events_to_voxel(const double* events, const int64_t* offsets, int B, int bins, int H, int W, float* voxel,
                       double* stats_scratch, int normalize, void* stream);

/* f-4: the voxel grids of the windows a reader cut out of an event stream
 * (data_readers/video_readers.py:118-141, 211-232): as cf_events_to_voxel plus event_preprocess's hot-pixel filter --
 * voxels with |v| > hot_pixel_threshold (the reference uses 25 / num_bins) are zeroed before the normalisation; <= 0: off. */
int cf_events_to_voxel_ex(const double* events, const int64_t* offsets, int B, int bins, int H, int W, float* voxel,
                          double* stats_scratch, int normalize, float hot_pixel_threshold, void* stream);

/* event_preprocess(grid, mode='std', filter_hot_pixel) (utils/event_process.py:193-216) of B grids that already sit on the
 * device ([B][voxels_per_grid] fp32, in place): zero |v| > hot_pixel_threshold (<= 0: off), then -- normalize != 0 -- mean 0 /
 * std 1 over each grid's non-zero voxels.  Used where the reference crops a grid before normalising it
 * (data_readers/MVSEC.py:389-403).  stats_scratch: 3*B doubles (may be NULL when normalize == 0). */
int cf_voxel_preprocess(float* voxel, int B, long long voxels_per_grid, double* stats_scratch, int normalize,
                        float hot_pixel_threshold, void* stream);

/* f-2  output stage, `np.uint8(pred_image * 255.)` (test_with_flow.py:174): fp32 product, truncation toward zero,
 * on the device (img: n floats in [0,1], out: n bytes).  Stateless, asynchronous on `stream`.  The PNG encoder stays on the
 * host (cista_flow_amd/utils/data_io.py: ImageWriter / FlowWriter on PIL); the flow colour coding is cf_flow_to_bgr below. */
int cf_quantize_u8(const float* img, unsigned char* out, long long n, void* stream);
/* f-2, the other half of the output stage: FlowWriter's colour coding `merge_optical_flow` (utils/data_io.py:9-29; caller
 * test_with_flow.py:178): flow [B][2][H][W] fp32 -> BGR uint8 [B][H][W][3] (the array cv2.imwrite receives), per image
 * H = uint8(angle * 180 / pi / 2), S = 255, V = uint8(255 * |flow| / max |flow|), then OpenCV's 8-bit HSV -> BGR.  scratch: B unsigned
 * ints on the device.  UNPINNED by the reference: cv2 is not installed where the goldens are generated, so this follows OpenCV's
 * published arithmetic and is tested against a numpy restatement of it (oracle.merge_optical_flow), not against cv2 itself. */
int cf_flow_to_bgr(const float* flow, int B, int H, int W, unsigned char* out_bgr, unsigned int* scratch, void* stream);


/* f-3  evaluation metrics on the device (loss.py of the reference).  Stateless and asynchronous on
 * `stream`; every result is written to DEVICE doubles (`out*`), so a caller reads a frame's scores with one small copy
 * whenever it wants them.  scratch: cf_metrics_scratch_doubles() device doubles.  Deterministic (fixed-order fp64 folds).
 *   cf_metrics_recon  out2 = {mse, psnr}: nn.MSELoss and PSNR(data_range=1) of ReconLoss.evaluate (loss.py:15-24,316-328;
 *                     psnr = 100 when mse < 1e-10).
 *   cf_metrics_ssim   out2 = {ssim, cs}: ReconLoss.evaluate's 'ssim' = pytorch_msssim.SSIM(data_range=1, size_average=True,
 *                     channel=1) (loss.py:314,319): 11-tap gaussian window (sigma 1.5), 'valid' convolution, K = (0.01, 0.03),
 *                     mean over all window positions of all `planes` = B*C images of H x W (H, W >= 11).  The package is not
 *                     installed offline: this follows its published algorithm (UNPINNED by the reference).  LPIPS needs
 *                     network weights (lpips + torchvision): not built.
 *   cf_metrics_flow   out6 = {photo_loss, epe, 1px, 3px, 5px, out}: FlowL1LossDict.evaluate (loss.py:237-265).  flow,
 *                     gt_flow NCHW [B][2][H][W]; gt_img0/1 [B][1][H][W]; flow_valid [B][1][H][W] or NULL (then
 *                     exp(-50*(warp(gt_img0, gt_flow) - gt_img1)^2), loss.py:241); warp_mode = the FrameWarp mode;
 *                     max_flow = 400 (loss.py:124).  epe / mag uses each pixel's own gt magnitude (the reference's
 *                     expression only runs at batch 1, where this is the same thing).
 *   cf_metrics_fwl    out3 = {var(sum_i warp_i(voxel_i; flow)), the same for zero flow, their ratio = FWL}
 *                     (voxel_warping_flow_loss, loss.py:27-83; ratio as in test_wo_flow.py:161).  voxel [B][C][H][W]. */
size_t cf_metrics_scratch_doubles(void);
int cf_metrics_recon(const float* rec, const float* target, long long n, double* out2, double* scratch, void* stream);
int cf_metrics_flow(const float* flow, const float* gt_flow, const float* gt_img0, const float* gt_img1,
                    const float* flow_valid, int B, int H, int W, int warp_mode, float max_flow, double* out6,
                    double* scratch, void* stream);
int cf_metrics_fwl(const float* voxel, const float* flow, int B, int C, int H, int W, double* out3, double* scratch,
                   void* stream);
int cf_metrics_ssim(const float* rec, const float* target, int planes, int H, int W, double* out2, double* scratch, void* stream);

/* measurement: when enabled, EVERY kernel launch of the fused paths (convolutions and the HBM-class kernels: warp,
 * up-sampling, InstanceNorm apply, correlation lookup / pyramid, flow up-sampling ...) is bracketed by HIP events on
 * the launch stream and the library's side streams are folded into the caller's stream (kernels run one at a time,
 * with the same grids as in normal operation, so a duration is that kernel alone on the chip -- what a roofline
 * fraction needs).  cf_profile_read synchronises them and returns, for conv tile kind t >= 1 (index 0 = all
 * contraction launches), the summed launch duration in ms, the summed algorithmic flops (2*M*N*K with the un-padded
 * K) and the launch count, then clears the records.  cf_conv_tile_name(t) = kernel symbol.
 * cf_profile_report_json: one row per (layer tag, kernel symbol, grid in work-items = rocprofv3's Grid_Size) over both
 * classes: "mfma" rows carry algorithmic flops, "hbm" rows algorithmic bytes (each input read once, each output
 * written once) in `work`.  Environment CF_SERIAL=1 gives the same serialised launch order without the events (the
 * mode profiles/<round>_ktrace_serial.txt is taken in). */
int cf_profile_enable(cf_handle* h, int on);
int cf_profile_read(cf_handle* h, double* ms, double* flops, long long* count, int n);
const char* cf_conv_tile_name(int tile);
/* executed / algorithmic matrix-core flops of that tile kind's kernel (4/9 Winograd F(2x2,3x3), 0.6 F(2,5), 0.25 F(4x4,3x3), 1 direct):
 * the rows of cf_profile_report_json carry it as "mfma_ratio" next to "tile", so the executed-MFMA roofline fraction is priced per
 * launch site from the library's own table */
double cf_conv_tile_mfma_ratio(int tile);

/* kernel-selection plan.  cf_plan_enable(h, 1): every convolution launch of the handle records (deduplicated) the integer fields of its
 * descriptor that the launcher's tile choice can depend on + the tile it took; cf_plan_json returns
 * {"fields": [names], "rows": [{"tag", "tile", "kernel", "desc": [ints in `fields` order]}]}.  cf_conv_plan replays ONE such descriptor
 * through the same chooser without launching anything (no GPU needed): tests/test_kernel_selection_cpu.py holds every BASELINE config's
 * committed table (tests/golden/kernel_selection.json, tools/gen_kernel_table.py) against it, so a launcher-heuristic change is a diff. */
int cf_plan_enable(cf_handle* h, int on);
const char* cf_plan_json(cf_handle* h);
int cf_conv_plan(const int* desc, int n, int* tile_out);
/* per-layer text table of the last cf_profile_read (layer, tile kind, launches, ms, TFLOP/s) */
const char* cf_profile_report(const cf_handle* h);
const char* cf_profile_report_json(const cf_handle* h);

/* single-operator entry points (used by the parity tests; same kernels as the fused paths) ------ */
/* conv2d on NHWC input (a_mode 0), fused x2-upsample input (a_mode 1) or planar NCHW small-Cin input
 * (a_mode 2); weight OIHW; out NHWC [B][Ho][Wo][Cout].  epi: 0 none 1 relu 2 sigmoid 3 tanh. */
int cf_op_conv2d(const float* in, int B, int Cin, int H, int W, const float* weight, const float* bias, int Cout,
                 int KH, int KW, int stride, int padT, int padL, int pad_mode, int a_mode, int epi, int tile,
                 float* out, void* stream);
/* tuning tool: the same conv launched `iters` times between two HIP events; *ms_out = avg launch ms */
int cf_op_conv2d_bench(const float* in, int B, int Cin, int H, int W, const float* weight, const float* bias, int Cout,
                       int KH, int KW, int stride, int padT, int padL, int pad_mode, int a_mode, int epi, int tile,
                       float* out, void* stream, int iters, float* ms_out, int precision);
int cf_op_instance_norm_relu(const float* x_nhwc, float* out_nhwc, int B, int C, int H, int W, float eps,
                             void* stream);
/* conv (no activation) whose epilogue also accumulates the InstanceNorm2d statistics of its output
   (raft_encoder.py:32-36: norm follows every conv of the 'instance' encoders); stats_out [B][Cout][2] =
   {mean, 1/sqrt(biased var + eps)} as inorm_apply consumes them.  a_mode / tile as cf_op_conv2d. */
int cf_op_conv2d_inorm_stats(const float* in, int B, int Cin, int H, int W, const float* weight, const float* bias,
                             int Cout, int KH, int KW, int stride, int padT, int padL, int pad_mode, int a_mode,
                             int tile, float* out, float* stats_out, float eps, void* stream);
/* all-pairs correlation + pyramid + lookup (a9/a10): fmaps NHWC [B][h][w][D], coords NCHW [B][2][h][w];
 * out NHWC [B][h][w][4*81] */
int cf_op_corr_lookup(const float* fmap1, const float* fmap2, const float* coords, float* out, int B, int D, int h,
                      int w, void* stream);
int cf_op_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream);
int cf_op_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CISTAFLOW_H */
