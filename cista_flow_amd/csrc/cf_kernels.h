// cf_kernels.h -- internal launcher interface of libcistaflow (gfx950 only).
//
// Data layout in HBM: every activation is fp32 NHWC ("pixel-major"): element
// (b, y, x, c) lives at  base + b*bs + (y*W + x)*ld + c , where ld >= C lets a
// producer write straight into a channel slice of a wider concat buffer (so
// torch.cat never materialises).  Boundary tensors with C <= 5 (event voxel,
// image, flow) stay planar NCHW exactly as the reference hands them over.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cf {

// What the last launcher on this thread put on its stream (kernel symbol, total work-items = rocprofv3's
// Grid_Size): the per-launch profiler of cf_api.hip reads it so that its HIP-event table can be joined with a
// rocprofv3 --kernel-trace of the same run by (kernel, grid).
struct LaunchInfo { const char* kernel; long threads; };
extern thread_local LaunchInfo g_last_launch;
inline void note_launch(const char* kernel, dim3 g, dim3 b) {
    g_last_launch.kernel = kernel;
    g_last_launch.threads = (long)g.x * g.y * g.z * b.x * b.y * b.z;
}

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//   M = output pixels of one image, N = output channels, K = taps * Cin.
// ---------------------------------------------------------------------------
enum AMode {
    A_NHWC = 0,      // up to 3 NHWC channel segments (zero-copy torch.cat)
    A_UPS2X = 1,     // NHWC source at (Hsrc,Wsrc) read through a fused bilinear x2
                     // upsample (align_corners=False); conv sees (Hin,Win)=(2Hsrc,2Wsrc)
    A_GATHER = 2,    // planar NCHW source with tiny Cin: K = taps*Cin flattened
};

enum Epi {
    EPI_NONE = 0,            // v = acc + bias
    EPI_RELU = 1,
    EPI_SIGMOID = 2,
    EPI_TANH = 3,
    EPI_SUB_FROM_AUX = 4,    // out = aux0 - v                         (ISTA: x1 - D(z))
    EPI_ADD_AUX_SHRINK = 5,  // out = softshrink(v + aux0, lam[n])     (ISTA: P(.) + z)
    EPI_RELU_ADD_AUX = 6,    // out = relu(v) + aux0                   (EIFusion)
    EPI_RELU_ADD_AUX_RELU = 7, // out = relu(aux0 + relu(v))           (BN residual block)
    EPI_LSTC = 8,            // ConvLSTC tail: o = sig(v); c = f*cprev + i*z0; out = o*tanh(c); out2 = c
    EPI_GRU_ZR = 9,          // n < split: out = sig(v) ; else out2[n-split] = sig(v)*aux0[n-split]
    EPI_GRU_Q = 10,          // q = tanh(v); out = (1-aux0)*aux1 + aux0*q
    EPI_TANH_RELU_SPLIT = 11,// n < split: out = tanh(v) ; else out2[n-split] = relu(v)
    EPI_SCALE = 12,          // out = acc * scale   (no bias)          (all-pairs correlation)
    EPI_LSTM_ACT = 13,       // n < split: sigmoid ; else tanh         (ConvLSTM gate pre-activation)
    EPI_ADD_AUX = 14,        // out = v + aux0                         (coords1 += delta_flow)
    EPI_BIAS_SCALE = 15,     // out = (acc + bias) * scale             (ERAFT: .25 * mask(net))
    EPI_LSTM_CELL = 16,      // ConvLSTM tail on gate-interleaved rows (quad = in | remember | out | cell of one hidden
                             // channel): c = sig(f)*aux0 + sig(i)*tanh(g); out = sig(o)*tanh(c); out2 = c   (base_layers.py:117-132)
};

struct ConvParams {
    // ---- A operand (activations) ----
    const float* in[3];
    int  seg_c[3];      // channels per segment (multiples of 16 in A_NHWC/A_UPS2X)
    int  seg_ld[3];     // pixel stride in floats
    long seg_bs[3];     // batch stride in floats
    int  nseg;
    int  Hin, Win;      // spatial size the convolution sees
    int  Hsrc, Wsrc;    // physical source size (A_UPS2X: half res; A_GATHER: un-padded image)
    int  Ho, Wo;
    int  KH, KW, stride, padT, padL;
    int  pad_mode;      // 0 zeros, 1 reflect
    int  a_mode;
    // A_GATHER extras: source is [B][g_cin][Hsrc][Wsrc]; the conv's virtual input is that image
    // shifted by (g_offy,g_offx) (ImagePadder zero pad on top/left), v' = g_scale*v + g_shift
    // inside the un-padded area, 0 outside.  g_subgrid: subtract the pixel grid (x for c=0,
    // y for c=1) so that coords1 is read as flow = coords1 - coords0.
    int   g_cin, g_offy, g_offx, g_subgrid;
    float g_scale, g_shift;
    // ---- B operand (packed weights [w_rows][Ktot], K contiguous) ----
    const float* w;
    const float* w_wino; // Winograd transform of w (nullable): F(2x2,3x3) of a 3x3 matrix (launch_wino_weights, tile 40) or F(2,5) of a 1x5 / 5x1 one (launch_wino1d_weights, tile 46)
    long wino_gs;       // floats between the Winograd matrices of consecutive weight groups (w_div)
    const float* w_wino4; // Winograd F(4x4,3x3) transform of w (launch_wino4_weights; nullable): enables tile 42
    long wino4_gs;
    const float* w_wino16; // F(2x2,3x3) transform of w in conv_wino16_kernel's order (launch_wino16_weights; nullable): enables tile 47
    long wino16_gs;
    const void* w16;    // f16 split copy of w (nullable; f16 modes fall back to splitting B while staging)
    long w_bs;          // weight stride between image groups (0 for ordinary weights; N*D for the correlation GEMM)
    int  w_div;         // images per weight group (<= 1: one matrix per image when w_bs != 0).  Two networks with the
                        // same layer shapes run as ONE launch over a 2B batch: images [0,B) use matrix 0, [B,2B) matrix 1
    long bias_gs;       // bias stride between the same groups (floats)
    int  w_rows;        // valid rows in the packed matrix
    int  Ktot;          // taps * cin_pad (A_NHWC/A_UPS2X) or round16(taps*Cin) (A_GATHER)
    int  cin_pad;       // per-tap K (multiple of 16); unused for A_GATHER
    const float* bias;  // [>= cout] or nullptr
    // ---- output ----
    float* out;  int out_ld;  long out_bs;  int out_cs;   // out[b*bs + m*ld + n*cs]
    float* out2; int out2_ld; long out2_bs;
    int  cout;          // store mask: n < cout
    int  epi;
    int  split;
    float scale;
    const float* aux0; int aux0_ld; long aux0_bs; int aux0_cs;
    const float* aux1; int aux1_ld; long aux1_bs;
    const float* aux2; int aux2_ld; long aux2_bs;
    const float* aux3; int aux3_ld; long aux3_bs;
    const float* lam;   // [cout] soft-threshold
    const float* addend; int addend_ld; long addend_bs;   // v += addend[b][m][n] before the epilogue op
    int  k_real;        // un-padded K (taps * Cin): algorithmic-flop bookkeeping only
    const char* tag;    // layer name for the profiler (host side only)
    int  prec;          // 0 fp32 MFMA, 3 f16x3 split MFMA, 1 plain f16 MFMA (w must point at the matching copy)
    int  b_f32;         // f16 modes: B operand is fp32 activations (correlation GEMM), split while staging
    int  sched;         // workgroup->tile map: -1 default (env CF_SCHED, else 1), 0 m-fastest, 1 XCD-aware
    // fused InstanceNorm statistics (EPI_NONE only): every 32-pixel x cout patch of the output adds its fp64
    // {sum, sum of squares} of v = acc + bias to st_partial[b][ceil(M/32)][cout][2]; launch_inorm_final folds them
    double* st_partial;
    long long* stamp;   // -DCF_STAMP builds only (tools/stamp_probe.py): per-wave cycle stamps of conv_dma_kernel
    int  tile_batch;    // > 0: choose the tile as if the batch were this large (keeps the arithmetic order of a launch that
                        // covers only part of a batch identical to the full-batch launch: ERAFT feature reuse)
    int  epi_vec;       // set by launch_conv: out / out2 / aux / addend rows are 16-byte aligned (dwordx4 tail)
};

// tile: 0 auto, else explicit (see conv_igemm.hip); tile_used (nullable) returns the choice
hipError_t launch_conv(const ConvParams& p, int batch, hipStream_t s, int tile = 0, int* tile_used = nullptr, bool dry = false);
const char* conv_tile_name(int tile);
double conv_tile_mfma_ratio(int tile);      // executed / algorithmic flops of that tile's kernel (Winograd forms < 1)
// U = G g G^T of a packed 3x3 matrix [rows][9][cin_pad] in conv_wino_kernel's block layout (wino_weight_floats floats)
hipError_t launch_wino_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s);
long wino_weight_floats(int rows, int cin_pad);
// the same for F(4x4,3x3) (conv_wino4_kernel's [n-block][chunk][36 pos][32 n][8 k] layout)
hipError_t launch_wino4_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s);
long wino4_weight_floats(int rows, int cin_pad);
// conv_wino_sk.hip, reached through launch_conv (tiles 44 / 45): conv_wino_kernel with the channel chunks split over sk = 2 / 4 wave groups
hipError_t launch_wino_sk(const ConvParams& p, int batch, hipStream_t s, int sk);
// conv_wino1d.hip, reached through launch_conv (tile 46): one-dimensional Winograd F(2,5) for 1x5 / 5x1 convolutions (the separable GRU);
// its transformed weights travel in w_wino / wino_gs (a 1x5 / 5x1 layer has no F(2x2,3x3) form, so the field is free)
hipError_t launch_wino1d_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s);
long wino1d_weight_floats(int rows, int cin_pad);
bool wino1d_ok(const ConvParams& p);
long wino1d_workgroups(const ConvParams& p, long batch);
hipError_t launch_wino1d(const ConvParams& p, int batch, hipStream_t s);
// conv_wino16.hip, reached through launch_conv (tile 47): F(2x2,3x3) in half-size workgroups (16 tiles x 32 channels, 16x16x4 MFMA)
hipError_t launch_wino16_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s);
long wino16_weight_floats(int rows, int cin_pad);
bool wino16_ok(const ConvParams& p);
int wino16_regions(int Ho, int Wo);
hipError_t launch_wino16(const ConvParams& p, int batch, hipStream_t s, bool deep = false);    // deep: conv_wino16_kernel<1> (tile 50)
// conv_wino_p.hip, reached through launch_conv (tile 48): conv_wino_kernel's arithmetic in persistent workgroups that walk several regions
bool wino_p_ok(const ConvParams& p);
int wino_p_walkers(const ConvParams& p, long NR, int pipe = 0);
hipError_t launch_wino_p(const ConvParams& p, int batch, hipStream_t s, int pipe = 0);      // pipe = 1: tile 49, software-pipelined chunk loop
long wino16_max();         // CF_WINO16_MAX (default 640; 0 = never take tile 47 and do not build its weights)
extern long g_wino4_min;   // launches with at least this many F(4x4,3x3) workgroups take tile 42 (0 = never; CF_WINO4_MIN at cf_create)
// conv_wino4.hip, reached through launch_conv (tile 42)
bool wino4_ok(const ConvParams& p);
int wino4_regions(int Ho, int Wo);
hipError_t launch_wino4(const ConvParams& p, int batch, hipStream_t s);
// number of statistics partials per image a convolution with st_partial writes (tile = the tile launch_conv used)
int conv_stats_chunks(const ConvParams& p, int tile);
// f16 split copy of a packed weight matrix (same byte size, LDS chunk format [16 hi | 16 lo])
hipError_t launch_split_weight_f16(const float* src, void* dst, long rows, int Ktot, hipStream_t s);

// ---------------------------------------------------------------------------
// Weight packing (device side, runs once per load_state_dict)
// ---------------------------------------------------------------------------
// src: OIHW [Cout][Cin][KH][KW]  ->  dst rows [row0 .. row0+Cout) of [rows][Ktot]
//   gather==0: k = tap*cin_pad + c ; gather==1: k = tap*Cin + c
// optional BatchNorm fold (eval): w' = w*g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps)+beta
// c_begin/c_count select an input-channel slice of src (c_count <= 0: all), written at channel dst_coff;
// accum != 0 adds to what is already packed (sums input channels that always carry identical data).
hipError_t launch_pack_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW,
                              int cin_pad, int Ktot, int row0, int gather, int c_begin, int c_count, int dst_coff,
                              int accum,
                              const float* bn_w, const float* bn_b, const float* bn_mean,
                              const float* bn_var, float bn_eps,
                              const float* bias_src, float* bias_dst, hipStream_t s, int interleave = 0);
// interleave = G > 1: output channel g*(Cout/G) + j is stored as row j*G + g (gate-interleaved rows, EPI_LSTM_CELL)

// ---------------------------------------------------------------------------
// HBM-bound kernels
// ---------------------------------------------------------------------------
// a4: forward/backward flow warp (grid_sample bilinear, align_corners=True, reflection,
// x normalised by W not W-1).  img/out: [B][H*W][C] with pixel stride ld.  flow is planar
// [B][2][Hf][Wf]; when (Hf,Wf) != (H,W) it is first resampled to (H,W) with
// interpolate(bilinear, align_corners=True) WITHOUT rescaling its values (e2v_model.py:190).
// flag (nullable): device int; when *flag == 0 the kernel copies img -> out (".any()" false).
hipError_t launch_warp(const float* img, int img_ld, long img_bs, const float* flow, int Hf, int Wf,
                       float* out, int out_ld, long out_bs, int B, int C, int H, int W,
                       int backward, const int* flag, hipStream_t s);
// the image warp and the sparse-code warp of one frame (e2v_model.py:186-191) in ONE launch; img2 == nullptr: one tensor
hipError_t launch_warp2(const float* img, int img_ld, long img_bs, float* out, int out_ld, long out_bs, int C, int H, int W,
                        const float* img2, int img2_ld, long img2_bs, float* out2, int out2_ld, long out2_bs, int C2, int H2,
                        int W2, const float* flow, int Hf, int Wf, int B, int backward, const int* flag, hipStream_t s);
// flag = any(flow != 0)
hipError_t launch_any_nonzero(const float* x, long n, int* flag, hipStream_t s);

// InstanceNorm2d(affine=False, eps): two-stage statistics + apply.
//   stats[b][c] = {mean, rstd}
// fold partial sums [B][nchunk][C][2] (launch_inorm_stats' own, or ConvParams::st_partial with nchunk = ceil(HW/32))
// into stats [B][C][2] = {mean, 1/sqrt(var + eps)} (biased variance)
hipError_t launch_inorm_final(const double* partial, int nchunk, int B, int HW, int C, float eps, float* stats,
                              hipStream_t s);
hipError_t launch_inorm_stats(const float* x, int ld, long bs, int B, int HW, int C, float eps,
                              double* partial, float* stats, hipStream_t s);
// out = relu(norm(x))                                   (res == nullptr)
// out = relu(res' + relu(norm(x))), res' = res or norm(res) when res_stats != nullptr
hipError_t launch_inorm_apply(const float* x, int ld, long bs, const float* stats,
                              const float* res, int res_ld, long res_bs, const float* res_stats,
                              float* out, int out_ld, long out_bs, int B, int HW, int C, hipStream_t s);

// correlation pyramid: dst[b][i][y][x] = avg 2x2 of src
hipError_t launch_corr_pool(const float* src, float* dst, long rows, int Hs, int Ws, hipStream_t s);
// correlation lookup (a10): out[b][i][lvl*81 + a*9 + bb], channels [324, out_ld) zeroed.
// also writes flow = coords1 - coords0 into motion[b][i][mo_off + {0,1}] when motion != nullptr.
struct LookupParams {
    const float* lvl[4]; int lh[4], lw[4];
    const float* coords1;       // planar [B][2][h8][w8]
    float* out; int out_ld;     // nullptr: no lookup (the flow-head step alone, after the last iteration)
    float* motion; int mo_ld; int mo_off;
    int B, h8, w8, radius, nlevels;
    // optional fused step in front of the lookup: coords1 += FlowHead.conv2(fh) (3x3, 256 -> 2, zero pad) at every query
    const float* fh; int fh_ld;             // nullptr: lookup at coords1 as it is; NHWC [B][N][256]
    const float* fh_w; int fh_ktot;         // packed rows 0 / 1 of [rows][9 * 256] (k = tap * 256 + c)
    const float* fh_bias;
    float* coords_out;                      // = coords1 (in place: a query only reads and writes its own pixel)
};
hipError_t launch_corr_lookup(const LookupParams& p, hipStream_t s);
// levels 1..3 of the correlation pyramid in one launch (bit-identical to three launch_corr_pool calls) + optionally coords1 = grid
// (+ flow_init) and *flag = 0 in the same launch (coords1 / flag nullable); corr_pyramid_lds_bytes > 64 KiB: not available, use the cascade
long corr_pyramid_lds_bytes(int H0, int W0);
hipError_t launch_corr_pyramid(const float* l0, float* l1, float* l2, float* l3, long rows, int H0, int W0, float* coords1,
                               const float* flow_init, int B, int h8, int w8, int* flag, hipStream_t s);

// coords1 = grid (+ flow_init)
hipError_t launch_coords_init(float* coords1, const float* flow_init, int B, int h8, int w8, hipStream_t s);
// flow_up = ds * interpolate(coords1 - coords0, x ds, bilinear, align_corners=True)   (padded, nullable)
// flow_final = unpad(flow_up)                                                          (nullable)
// flag (nullable) |= any(flow_final != 0)
// flow_low (nullable): additionally coords1 - coords0 on the 1/8 grid, by trailing blocks of the same launch
hipError_t launch_upflow(const float* coords1, int B, int h8, int w8, int ds, float* flow_up,
                         float* flow_final, int H, int W, int padH, int padW, int* flag, hipStream_t s, float* flow_low = nullptr);

// learned convex x8 up-sampling (ERAFT/eraft.py:77-88, idn/idedeq.py:48-61): softmax over the 9 neighbours of
// mask [B][N][576] (channel k*64 + i*8 + j) applied to unfold(8 * flow, 3x3, pad 1); flow = coords1 - coords0
// when `coords_is_flow == 0`, else `coords1` already holds the flow.  Writes the padded flow_up (nullable) and
// the un-padded flow_final (nullable), raises `flag` on any non-zero flow_final value.
// add (nullable, padded [B][2][8h8][8w8]): flow_final / total_out receive add + up (IDNet: flow_init + delta).
hipError_t launch_convex_upsample(const float* coords1, int coords_is_flow, const float* mask, int mask_ld, int B,
                                  int h8, int w8, float* flow_up, float* flow_final, int H, int W, int padH,
                                  int padW, int* flag, const float* add, float* total_out, hipStream_t s);

// IDNet deblur (idn/idedeq.py:74-92): out[b][t] = grid_sample(bins[b][t], p + flow*t/(T-1)) with the grid
// normalised by (W-1) but align_corners=False and zero padding (sampled pixel x*W/(W-1) - 0.5: not an identity
// at zero flow).  bins: un-padded [B][T][H][W]; flow (nullable = 0): padded [B][2][Hp][Wp]; out: [B][T][Hp][Wp].
hipError_t launch_idn_deblur(const float* bins, const float* flow, float* out, int B, int T, int H, int W, int padH,
                             int padW, hipStream_t s);

// f-1: events -> temporal-bilinear voxel grid + non-zero mean/std normalisation
// (utils/event_process.py:15-72,193-216).  events: [total][4] fp64 rows (t, x, y, polarity), sequences
// concatenated, offsets[B+1]; voxel: [B][bins][H][W] fp32 (zeroed here); stats: [B][3] fp64 scratch.
// hot > 0: voxels with |v| > hot are zeroed before the normalisation (event_preprocess(filter_hot_pixel=True): 25 / bins)
hipError_t launch_events_to_voxel(const double* events, const long* offsets, int B, int bins, int H, int W,
                                  float* voxel, double* stats, int normalize, hipStream_t s, float hot = 0.f);

// event_preprocess('std', filter_hot_pixel) of B device-resident grids of per_seq voxels each, in place
hipError_t launch_voxel_preprocess(float* voxel, int B, long per_seq, double* stats, int normalize, float hot, hipStream_t s);

// layout helpers for the Python boundary / tests
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int dst_ld, int B, int C, int HW, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* src, int src_ld, float* dst, int B, int C, int HW, hipStream_t s);

// F.interpolate(x2, bilinear, align_corners=False) of an NHWC tensor [B][Hs][Ws][C] -> [B][2Hs][2Ws][C]
hipError_t launch_upsample2x_nhwc(const float* src, int s_ld, long s_bs, float* dst, int d_ld, long d_bs, int B, int Hs, int Ws,
                                  int C, hipStream_t s);

// f-2: np.uint8(pred * 255.) of the reconstructed frames (test_with_flow.py:174)
hipError_t launch_quantize_u8(const float* x, unsigned char* out, long n, hipStream_t s);
// f-2: FlowWriter's colour coding (utils/data_io.py:9-29): flow [B][2][H][W] -> BGR uint8 [B][H][W][3]; scratch: B unsigned ints.
// UNPINNED by the reference (cv2 absent): OpenCV's published 8-bit HSV -> BGR arithmetic, see pointwise.hip
hipError_t launch_flow_to_bgr(const float* flow, int B, int H, int W, unsigned char* out, unsigned* scratch, hipStream_t s);

// f-3: evaluation metrics on the device (metrics.hip).  out / scratch are device doubles; scratch holds
// metrics_scratch_doubles() entries.  All asynchronous on `s`, deterministic (fixed-order fp64 folds).
long metrics_scratch_doubles();
// out[2] = {mse, psnr}   (loss.py:15-24, nn.MSELoss)
hipError_t launch_metrics_recon(const float* rec, const float* tgt, long n, double* out, double* scratch, hipStream_t s);
// out[6] = {photo_loss, epe, 1px, 3px, 5px, out}   (FlowL1LossDict.evaluate, loss.py:237-265); valid nullable
hipError_t launch_metrics_flow(const float* flow, const float* gt, const float* img0, const float* img1, const float* valid,
                               int B, int H, int W, int backward, float max_flow, double* out, double* scratch, hipStream_t s);
// out[3] = {var(warped events | flow), var(warped events | zero flow), ratio = FWL}   (loss.py:27-83, test_wo_flow.py:161)
hipError_t launch_metrics_fwl(const float* voxel, const float* flow, int B, int C, int H, int W, double* out, double* scratch,
                              hipStream_t s);

// out[2] = {ssim, cs}: pytorch_msssim.SSIM(data_range=1, win 11, sigma 1.5, K (0.01, 0.03), size_average) over `planes` images of
// H x W (loss.py:314,319); H, W >= 11
hipError_t launch_metrics_ssim(const float* x, const float* y, int planes, int H, int W, double* out, double* scratch, hipStream_t s);

}  // namespace cf
