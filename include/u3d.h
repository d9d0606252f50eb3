/*
 * u3d.h — C-ABI of the MI355X-native (gfx950) 3D U-Net forward/backward hot path.
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  The reference (wolny/pytorch-3dunet 1.9.6) has
 * NO FFI of its own: its hot path bottoms out in torch.nn modules that dispatch into ATen.  Every
 * entry point below therefore cites the reference call site (file:line under /root/reference) whose
 * ATen operator it replaces.  A maintainer of the reference binds these with ctypes (see
 * INTEGRATION.md for the stub); our host-side mirror is pytorch-3dunet_amd/pytorch3dunet_amd.
 *
 * Conventions
 *   - plain C, POD arguments only: raw device pointers, ints, a hipStream_t passed as void*.
 *   - every function returns 0 on success or a negative U3D_E* code; the message is available from
 *     u3d_last_error() (thread-local).  Nothing throws or aborts across the ABI.
 *   - kernels are enqueued asynchronously on `stream`; the library never synchronises, never
 *     allocates or frees user-visible memory and keeps no pointer beyond a call.
 *   - re-entrant and thread-safe (autograd worker threads, nn.DataParallel replica threads).
 *   - all activations are fp32, channels-last "NDHWC": elem(n,z,y,x,c) = ((n*D+z)*H+y)*W+x)*C + c.
 *     (For in_channels == out_channels == 1 this is byte-identical to the reference's NCDHW.)
 *   - N*D*H*W must be < 2^31.
 */
#ifndef U3D_H
#define U3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define U3D_VERSION 128 /* 128: u3d_conv3d_wgrad_bf16_job / _b16_job (the GroupNorm-backward reduction rides in the bf16 weight gradient's reduce launch too); 127: u3d_subpixel_conv_dgrad_reps, u3d_gn_bwd_job_t::reps_hi; 126: replica rows also from u3d_chan_stats_reps, u3d_conv3d_small_cin_fwd_reps, u3d_conv1x1_head_bwd_reps + u3d_cvt_f64_f32_sum; 125: replica rows of the statistics tables (u3d_conv3d_ex_reps, u3d_gn_finalize_reps, u3d_gn_bwd_job_t::reps_lo); 124: u3d_bce_dice_scratch_doubles (per-block partials instead of atomics); 123: u3d_conv3d_wgrad_job (the GroupNorm-backward reduction rides in the weight-gradient reduce launch); 122: round 6 — u3d_gn_finalize_split / u3d_gn_bwd_finalize_split (compact half tables of a virtual-concat layer), u3d_adam_step, u3d_chan_stats_children, u3d_pack_weights_batch_cells, tuning key 18; 121: u3d_convtr3d_fwd_t8_b16_ex; 120: flat 5 x 10 x 10 tile of the bf16-storage convolutions (u3d_conv3d_bf16_tile_variant planes = 5), 24 tuning keys; 119: round 5 — ragged volumes on the persistent kernels, u3d_conv3d_variant / u3d_conv3d_wgrad_variant; 112: BatchNorm / conv-bias / dropout entry points (u3d_norm.hip); 113: one-launch bf16 weight packing (u3d_pack_weights_bf16_batch), 16 tuning keys, bf16 activation storage (*_b16); 114: 1x1x1 convolution on the bf16 matrix pipe (u3d_conv1x1_*_mfma_b16); 115: round 4 — u3d_conv3d_bf16_tile_variant, tuning key 12 (free slots in the persistent grids); 116: u3d_conv3d_wgrad_bf16_b16_variant; 117: u3d_convtr3d_dgrad_t8*_ex (split-K); 118: u3d_se_*_b16 */

#define U3D_OK 0
#define U3D_EINVAL (-1)  /* bad shape / argument */
#define U3D_EHIP (-2)    /* a HIP runtime call failed */
#define U3D_EARCH (-3)   /* device is not gfx950 */
#define U3D_EWORKSPACE (-4) /* workspace too small */

typedef void* u3d_stream_t; /* hipStream_t */

/* A (possibly virtual) activation tensor read by the conv / stats / GN-backward kernels.
 *
 * Channels [0,C0) come from p0 at full resolution (N,D,H,W,C0).  Channels [C0,C0+C1) come from p1, a
 * LOW-resolution tensor (N,D1,H1,W1,C1) read through nearest-neighbour index maps — this is the
 * never-materialised `torch.cat((encoder_features, F.interpolate(x, size, mode="nearest")), dim=1)` of
 * buildingblocks.py:491 + :614 (skip channels first).  zmap/ymap/xmap are device int32 tables of length
 * D/H/W giving the source index (host computes them with PyTorch's float32 formula
 * min(floor(dst * (in/out)), in-1)).  C1 == 0 => plain tensor.
 *
 * affine, if non-NULL, is the fused GroupNorm apply of buildingblocks.py:75 in 'gcr' order: a device
 * table [N][C0+C1][2] of (a,b) with value = x*a + b, produced by u3d_gn_finalize.  Zero padding of the
 * convolution is applied AFTER the affine (padded taps contribute exactly 0), matching
 * nn.Conv3d(padding=1) applied to the GroupNorm output. */
typedef struct {
    const float* p0;
    const float* p1;
    const int32_t* zmap;
    const int32_t* ymap;
    const int32_t* xmap;
    const float* affine;
    int32_t C0, C1;
    int32_t D1, H1, W1;
} u3d_src_t;

/* ---- library ------------------------------------------------------------------------------- */
int u3d_version(void);
const char* u3d_last_error(void);
/* 0 if `device` is a gfx950 part, U3D_EARCH otherwise. */
int u3d_check_device(int device);
/* Process-wide performance knobs for A/B measurements (never change results).
 * key 0: forced N-tiles per block of u3d_conv3d (1,2,3; 0 = automatic); key 1: wgrad split override; key 12: block slots the
 * persistent convolution grids leave FREE (of 2 per CU) so that kernels of other streams — RCCL's gradient all-reduce,
 * parallel.py — find room beside them (it shrinks the fp32 persistent grids of u3d_conv3d ONLY: the bf16 kernels and every other
 * launch ignore it); the other keys: see the list at the top of csrc/u3d_conv.hip (24 slots; the environment
 * variable U3D_TUNE=key:value,... sets them at load time). */
int u3d_set_tuning(int key, int value);
/* Developer aid (tools/wave_timeline.py): while a device buffer is registered, u3d_conv3d launches an instrumented
 * twin of the kernel in which every wave records 24 int64 (block, HW_ID, XCC_ID, shader-clock stamps at entry /
 * first tile staged / end of each chunk's k-loop / epilogue start / exit).  NULL switches it off (default). */
int u3d_set_profile_buffer(void* device_buffer, size_t bytes);
/* Developer aid (tools/overlap_probe.py): `blocks` workgroups stream `n` floats in place `passes` times in 64 KiB pieces, paced
 * against the wall clock so that the launch lasts at least `min_seconds` (0 = unpaced) — the shape of a link-bound RCCL ring
 * all-reduce (a handful of channels moving data at xGMI rate), which a 1-rank process group cannot launch.  Values are unchanged. */
int u3d_debug_stream_pass(int device, u3d_stream_t stream, float* buf, long long n, int blocks, int passes, double min_seconds);
/* Measurement aid: 2 blocks per CU of nothing but v_mfma_f32_32x32x2_f32 on register operands (mode 0 zeros, 1 one constant pair, 2
 * random values per lane and step); *flop_out (host) = FLOPs executed.  Timed by bench.py: the fp32 matrix pipe's sustained rate depends
 * on the operand data (142 TFLOP/s on random operands, 155-158 on constants) — the ceiling of any roofline fraction on real activations. */
int u3d_debug_mfma_f32_rate(int device, u3d_stream_t stream, float* sink, int iters, int mode, double* flop_out);

/* ---- weight packing -------------------------------------------------------------------------
 * Reference weights stay nn.Parameters in (Cout,Cin,3,3,3) layout (checkpoint compatibility,
 * utils.py:36-65).  The MFMA kernels read a packed image [chunk][tap][s][ntile][lane][4]:
 *   mode 0 (forward):  B[k=(tap,c)][n=cout]      = w[cout][c][tap]
 *   mode 1 (dgrad):    B[k=(tap,cout)][n=c]      = w[cout][c][26-tap]   (flipped taps, swapped roles)
 * u3d_packed_weight_floats gives the size of the image in floats for (Cin,Cout,mode). */
size_t u3d_packed_weight_floats(int Cin, int Cout, int mode);
int u3d_pack_weights(int device, u3d_stream_t stream, const float* w, int Cout, int Cin, int mode, float* packed);
/* All layers of a model in ONE launch (weights change every optimizer step: 2 x 13 pack launches per step otherwise).
 * descs: DEVICE array of n descriptors sorted by `first` = cumulative float offset of the layer's image(s) within the
 * concatenation (first of descriptor 0 is 0); total_floats = sum of u3d_packed_weight_floats over the descriptors. */
/* A descriptor may also pack a CHANNEL SLICE of a weight: w points at input channel c_off of the parent
 * (parent + c_off*27), Cin is the slice width and cin_stride the parent's channel count (0 = Cin, a whole weight).
 * mode 2 packs the slice's sub-pixel image (u3d_subpixel_packed_floats(Cin, Cout) floats, see u3d_subpixel_conv_fwd),
 * mode 3 its data-gradient image (u3d_subpixel_dgrad_packed_floats(Cout, Cin) floats). */
typedef struct {
    const float* w; /* (Cout,Cin,3,3,3), or a channel slice of it */
    float* packed;  /* u3d_packed_weight_floats(Cin, Cout, mode) floats */
    int64_t first;
    int32_t Cout, Cin, mode, cin_stride;
} u3d_pack_desc_t;
int u3d_pack_weights_batch(int device, u3d_stream_t stream, const u3d_pack_desc_t* descs_device, int n,
                           int64_t total_floats);
/* The same images (modes 0..3, bit for bit) at memory rate (round 6): a block owns one (16-channel contraction chunk, 32-channel
 * n-tile) cell of one image — contiguous runs of the reference layout read with 16-byte loads, transposed through LDS, written 16 bytes
 * per lane — instead of one 4-byte gather at a stride of 27 floats per element.  descs: as above, but `first` = first BLOCK of the image
 * within the launch; an image takes u3d_pack_weights_cells_blocks(w, Cin, Cout, mode, cin_stride) blocks (its cells + one tail block;
 * 0 = not eligible — w not 16-byte aligned, or Cin / Cout / cin_stride not multiples of 4 — such an image stays on
 * u3d_pack_weights_batch); total_blocks = the sum over the descriptors. */
long long u3d_pack_weights_cells_blocks(const float* w, int Cin, int Cout, int mode, int cin_stride);
int u3d_pack_weights_batch_cells(int device, u3d_stream_t stream, const u3d_pack_desc_t* descs_device, int n, long long total_blocks);

/* ---- Conv3d 3x3x3, stride 1, pad 1, bias=False ------------------------------------------------
 * Replaces nn.Conv3d(in,out,3,padding=1,bias=False) (buildingblocks.py:56) forward, and — called with
 * mode-1 packed weights on dy — its data gradient (autograd of trainer.py:245).  Implicit GEMM on
 * v_mfma_f32_32x32x2_f32: M = 256-voxel tile (4x8x8), N = 32/64 output channels, K = 27*Cin, input halo
 * tile staged through LDS with the GroupNorm affine fused into the load.
 *
 *   out        (N,D,H,W,Cout) fp32
 *   relu       apply max(v,0) in the epilogue (nn.ReLU, buildingblocks.py:47)
 *   out_stats  optional device double[N][Cout][2]: += per-(n,channel) sum and sum of squares of the
 *              written outputs (feeds the next GroupNorm without re-reading the activation)
 *   gx/gstats  optional (dgrad use): gx is the layer's forward input (pre-GroupNorm, virtual concat
 *              allowed; its affine field is ignored); gstats double[N][Cout][2] += (sum dg, sum dg*x) —
 *              the two reductions GroupNorm backward needs.
 */
int u3d_conv3d(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out,
               int N, int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx,
               double* gstats);

/* The same convolution with a residual added before the ReLU: out = [relu](conv(src) + residual) — the tail of
 * ResNetBlock.forward (buildingblocks.py:277-288: conv3 without non-linearity, `out += residual`, non-linearity).
 * residual: (N,D,H,W,Cout) fp32. */
int u3d_conv3d_residual(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out,
                        int N, int D, int H, int W, int Cout, int relu, double* out_stats, const float* residual);

/* The general entry point: u3d_conv3d / u3d_conv3d_residual plus an optional scratch buffer.  At the bottom of the U
 * (few 4x8x8 tiles, many channels: 16 tiles x 4 channel blocks for 128 channels at 8x16x16) one block per (tile,
 * channel block) leaves most of the 256 CUs idle; given a workspace of u3d_conv3d_workspace_floats() floats the
 * reduction over input channels is split over several blocks whose partial sums are added in a fixed order by a second
 * kernel that also applies the epilogue (residual, ReLU, statistics) — same results contract as u3d_conv3d.
 * u3d_conv3d_workspace_floats() returns 0 for shapes that never split; workspace may be NULL (no split).
 * residual and gx/gstats are mutually exclusive. */
long long u3d_conv3d_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int u3d_conv3d_ex(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out, int N,
                  int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx, double* gstats,
                  const float* residual, float* workspace, long long workspace_floats);

/* u3d_conv3d_ex with a statistics table of stat_reps replica rows: out_stats / gstats are [stat_reps][N][Cout][2] doubles, zeroed by the
 * caller.  The persistent kernels' 512 blocks flush a sample's sums at the same time, and a same-address f64 atomic retires every 19.5 ns
 * (tools/atomic_bench.hip): 10 us behind the last tile of a launch on one row, 1.3 us on eight.  Block b adds to row b % stat_reps; every
 * other kernel variant adds to row 0.  The table the reference's GroupNorm sees (buildingblocks.py:70) is the sum over the rows:
 * u3d_gn_finalize_reps / u3d_gn_bwd_job_t::reps_lo take it in that form.  stat_reps = 1 is u3d_conv3d_ex. */
int u3d_conv3d_ex_reps(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out, int N,
                       int D, int H, int W, int Cout, int relu, double* out_stats, const u3d_src_t* gx, double* gstats,
                       const float* residual, float* workspace, long long workspace_floats, int stat_reps);

/* ---- 3x3x3 convolution over a nearest-2x-upsampled tensor, without the upsampled work ------------------------
 * The upsampled half of cat(skip, F.interpolate(low, nearest)) (buildingblocks.py:491,:614) feeding the decoder's first
 * Conv3d (buildingblocks.py:56): for an output voxel of parity p the three taps per dimension read only two low-res
 * voxels, so each of the 8 parity classes is a 2x2x2 convolution over the low-res grid with pre-summed weights — 8/27 of
 * the multiply-adds (csrc/u3d_subpix.hip).  Requires an exact 2x upsampling (output dims 2*D1, 2*H1, 2*W1), C1 % 4 ==
 * Cout % 4 == 0.
 *   u3d_pack_subpixel_weights: w is the full (Cout, Cin_total, 3,3,3) weight, channels [c_off, c_off + C1) are packed.
 *   u3d_subpixel_conv_fwd: low (N,D1,H1,W1,C1); affine optional GroupNorm (a,b) rows of those channels, sample n at
 *     affine + n * affine_sample_stride (floats) — a slice of the layer's [N][Ctot][2] table works in place;
 *     out (N,2*D1,2*H1,2*W1,Cout) receives the plain partial sums (no ReLU / statistics): add the skip half with
 *     u3d_conv3d_residual(skip, weights of the first C0 channels, residual = out). */
/*   u3d_subpixel_conv_dgrad: the data gradient with respect to the LOW-RES tensor, i.e. the reference's conv data gradient
 *     summed over the 8 children of every low-res voxel (the backward of F.interpolate(nearest)) in one pass: a 4x4x4-tap,
 *     stride-2 gather of dz with pre-summed taps.  dz (N,2*D1,2*H1,2*W1,Cout); dlow (N,D1,H1,W1,C1); optional gstats
 *     double[N][C1][2] += (sum dlow, sum dlow * x_low) — the GroupNorm-backward sums of those channels (equal to the
 *     full-resolution sums).  Packed image: u3d_pack_subpixel_dgrad_weights (batch descriptor mode 3). */
long long u3d_subpixel_packed_floats(int C1, int Cout);
long long u3d_subpixel_dgrad_packed_floats(int Cout, int C1);
int u3d_pack_subpixel_dgrad_weights(int device, u3d_stream_t stream, const float* w, int Cout, int Cin_total, int c_off,
                                    int C1, float* packed);
int u3d_subpixel_conv_dgrad(int device, u3d_stream_t stream, const float* dz, const float* packed, const float* x_low,
                            float* dlow, double* gstats, int N, int D1, int H1, int W1, int C1, int Cout);
/* ... with gstats as `reps` replica rows [reps][N][C1][2] (zeroed by the caller; see u3d_conv3d_ex_reps; u3d_gn_bwd_job_t::reps_hi). */
int u3d_subpixel_conv_dgrad_reps(int device, u3d_stream_t stream, const float* dz, const float* packed, const float* x_low, float* dlow,
                                 double* gstats, int N, int D1, int H1, int W1, int C1, int Cout, int reps);
int u3d_pack_subpixel_weights(int device, u3d_stream_t stream, const float* w, int Cout, int Cin_total, int c_off, int C1,
                              float* packed);
/*   workspace (optional, u3d_subpixel_fwd_workspace_floats() floats, 0 for shapes that never split): on small grids the
 *     channel reduction is split over several blocks whose partial sums are added in a fixed order. */
long long u3d_subpixel_fwd_workspace_floats(int N, int D1, int H1, int W1, int C1, int Cout);
int u3d_subpixel_conv_fwd(int device, u3d_stream_t stream, const float* low, const float* affine,
                          long long affine_sample_stride, const float* packed, float* out, int N, int D1, int H1, int W1,
                          int C1, int Cout, float* workspace, long long workspace_floats);

/* Weight gradient of the same convolution: dw[cout][cin][tap] = sum_{n,v} dz[n,v,cout] * g[n,v+tap,cin]
 * with g = src (GroupNorm affine fused on load, zero padded).  Split-K over voxel tiles with a
 * deterministic two-pass reduction.  workspace must hold u3d_wgrad_workspace_floats() floats.
 * dw is written (not accumulated) in the reference layout (Cout,Cin,3,3,3). */
size_t u3d_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int u3d_conv3d_wgrad(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw, int N,
                     int D, int H, int W, int Cout, float* workspace, size_t workspace_floats);

/* Host-only: which kernel variant u3d_conv3d[_ex|_residual] / u3d_conv3d_wgrad[_strided] run for a shape (16-byte aligned tensors,
 * channel counts that are multiples of 4).  src_kind: 0 = plain source, 1 = virtual source whose low-res half is an exact 2x
 * upsampling, 2 = virtual source through general index maps (odd sizes: F.interpolate to 2n+1, buildingblocks.py:598-614).
 * u3d_conv3d_variant: 0 = generic kernel (one block per tile, per-item bounds arithmetic), 1 = persistent kernel, every tile
 * inside the volume, 2 = persistent kernel with ragged last tiles (round 5: sizes that are not multiples of 4 x 8 x 8, e.g. the
 * reference's shipped 80 x 170 x 170 patch, resources/3DUnet_confocal_boundary/train_config.yml:94), 3 = split-K (has_workspace).
 * u3d_conv3d_wgrad_variant: 0 = generic staging, 1 / 2 = constant-offset staging without / with ragged last tiles, | 4 = tap pairs
 * (Cin <= 16).  No reference counterpart (ATen picks its algorithm behind buildingblocks.py:56); the parity tests assert with it
 * that a ragged shape runs the fast variants. */
int u3d_conv3d_variant(int N, int D, int H, int W, int Cin, int Cout, int src_kind, int has_workspace);
int u3d_conv3d_wgrad_variant(int N, int D, int H, int W, int Cin, int Cout, int src_kind);

/* Same, writing the gradient of a CHANNEL SLICE of a wider weight: dw points at the slice's first input channel inside the
 * (Cout, dw_cin_stride, 3,3,3) gradient; src holds only the slice's channels. */
int u3d_conv3d_wgrad_strided(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw,
                             int dw_cin_stride, int N, int D, int H, int W, int Cout, float* workspace,
                             size_t workspace_floats);

/* u3d_conv3d_wgrad_strided with the GroupNorm-backward reduction of the SAME layer's input (u3d_gn_bwd_finalize /
 * u3d_gn_bwd_finalize_split: one block) riding as one extra block of the launch that reduces the weight gradient: the caller
 * runs the data gradient (which produces gstats) first, then this.  Results are those of the two separate calls, bit for bit
 * (same device code, csrc/u3d_gn.h).  job == NULL: plain u3d_conv3d_wgrad_strided; dw_cin_stride == 0: the source's channel count.
 * u3d_conv3d_wgrad_job_supported(N, C, G): 1 when the reduction of N x C channels in G groups fits the block's LDS.
 * Reference: the two are separate autograd nodes (ConvolutionBackward0 / NativeGroupNormBackward0 behind buildingblocks.py:56,70). */
typedef struct u3d_gn_bwd_job {
    const double* gstats_lo; /* [N][C0][2] sums (sum dg, sum dg * x) of channels [0, C0) */
    const double* gstats_hi; /* [N][C1][2] sums of channels [C0, C0 + C1), or NULL (C1 == 0: one table) */
    const float* mean_rstd;  /* [N][G][2] from the forward finalize */
    const float* gamma;      /* [C] */
    float* dgamma;           /* [C] out */
    float* dbeta;            /* [C] out */
    float* coef;             /* [N][3][C] out: (p, q, r) of dx = p * dg + q * x + r */
    float* coef_hi;          /* [N][3][C1] out or NULL: (p, hi_scale * q, hi_scale * r) of the upper channels */
    double count;            /* voxels per sample */
    int32_t C0, C1, N, G;
    float hi_scale;
    int32_t reps_lo; /* 0 / 1: gstats_lo is one table; r > 1: r replica rows [r][N][C0][2] whose sum is the table (u3d_conv3d_ex_reps) */
    int32_t reps_hi; /* the same for gstats_hi (u3d_subpixel_conv_dgrad_reps) */
    int32_t reserved;
} u3d_gn_bwd_job_t;
int u3d_conv3d_wgrad_job_supported(int N, int C, int G);
int u3d_conv3d_wgrad_job(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw, int dw_cin_stride,
                         int N, int D, int H, int W, int Cout, float* workspace, size_t workspace_floats,
                         const u3d_gn_bwd_job_t* job);

/* Weight gradient of the upsampled half of a decoder's first convolution (see u3d_subpixel_conv_fwd): the 64 matrices
 * sum_j g_low[j + p - 1 + e] (x) dz[2j + p] (8 parity classes p x 8 tap halves e) over the low-res grid — 8/27 of the
 * multiply-adds of the reference formulation — folded into the 27 taps by a fixed-order reduce.  dw points at the first
 * upsampled input channel inside the (Cout, dw_cin_stride, 3,3,3) gradient.  low (N,D1,H1,W1,C1) with the optional
 * GroupNorm affine slice (as u3d_subpixel_conv_fwd); dz (N,2*D1,2*H1,2*W1,Cout). */
long long u3d_subpixel_wgrad_workspace_floats(int N, int D1, int H1, int W1, int C1, int Cout);
int u3d_subpixel_conv_wgrad(int device, u3d_stream_t stream, const float* low, const float* affine,
                            long long affine_sample_stride, const float* dz, float* dw, int dw_cin_stride, int N, int D1,
                            int H1, int W1, int C1, int Cout, float* workspace, long long workspace_floats);

/* ---- decoder levels that upsample n -> 2n + 1 along some axes (round 5) --------------------------------------------------
 * F.interpolate(x, size=skip, mode="nearest") to an ODD skip size (buildingblocks.py:598-614, :491 — the reference's shipped
 * 80 x 170 x 170 patch pools 85 -> 42 and upsamples 42 -> 85) reads src = (dst - 1) >> 1 for dst >= 1 and 0 for dst = 0: the exact
 * 2x upsampling shifted by one voxel with low[0] once more in front.  Every output voxel d >= 2 along such an axis sees exactly the
 * shifted 2x tensor under its three taps, so the three sub-pixel kernels run on a WINDOW — out[d] / dz[d] with d = u + o, u the voxel
 * of the 2x grid — and the slab d < 2 runs on the general kernel restricted to a box:
 *   win (forward)  = {Do, Ho, Wo, oz, oy, ox}: dims of `out`, shift o = 1 on an n -> 2n + 1 axis (0 on an exact one).  Voxels d = o on
 *                    a shifted axis are written but WRONG (they miss the extra copy of low[0]); u3d_conv3d_box overwrites the slab.
 *   win (gradients) = {Dd, Hd, Wd, oz, oy, ox, lz, ly, lx}: dims of dz, shift, and the first voxel u of the 2x grid that counts (1 on
 *                    a shifted axis: the outputs d >= 2).
 *   u3d_conv3d_box: plain convolution of `src` (C0 may be 0: only the upsampled half; p0 must still be readable) for the output
 *                    voxels inside out_box = {z0, y0, x0, z1, y1, x1}; in_mask = {mz, my, mx} or NULL: a source voxel counts only if
 *                    mz && z < mz || my && y < my || mx && x < mx (the data gradient of the slab: dz outside it is zero).
 *   u3d_conv3d_wgrad_box: weight gradient over the dz voxels inside `box` only (workspace: u3d_wgrad_workspace_floats of the box dims).
 *   u3d_nearest_childsum_add: dlow[j] += sum over the children of low-res cell j of dv (a full-resolution gradient that is valid in
 *                    the slab's dilation) for the cells with cz && jz < cz || cy && jy < cy || cx && jx < cx, + their share of the
 *                    GroupNorm-backward sums gstats[n][c] += (sum, sum * x_low).  zlo / ylo / xlo: first child of every low-res
 *                    index (n_in + 1 entries), as for u3d_gn_bwd_apply_up. */
int u3d_subpixel_conv_fwd_win(int device, u3d_stream_t stream, const float* low, const float* affine, long long affine_sample_stride,
                              const float* packed, float* out, int N, int D1, int H1, int W1, int C1, int Cout, const int* win);
int u3d_subpixel_conv_dgrad_win(int device, u3d_stream_t stream, const float* dz, const float* packed, const float* x_low, float* dlow,
                                double* gstats, int N, int D1, int H1, int W1, int C1, int Cout, const int* win);
int u3d_subpixel_conv_wgrad_win(int device, u3d_stream_t stream, const float* low, const float* affine, long long affine_sample_stride,
                                const float* dz, float* dw, int dw_cin_stride, int N, int D1, int H1, int W1, int C1, int Cout,
                                float* workspace, long long workspace_floats, const int* win);
int u3d_conv3d_box(int device, u3d_stream_t stream, const u3d_src_t* src, const float* packed_w, float* out, int N, int D, int H, int W,
                   int Cout, const int* out_box, const int* in_mask);
int u3d_conv3d_wgrad_box(int device, u3d_stream_t stream, const u3d_src_t* src, const float* dz, float* dw, int N, int D, int H, int W,
                         int Cout, float* workspace, size_t workspace_floats, const int* box);
/* GroupNorm backward on the low-res producer of such a level: out = (p * dlow + children * (q * x + r)) * [x > 0 if relu_mask] with
 * children = (2 + [ez && z == 0]) (2 + [ey && y == 0]) (2 + [ex && x == 0]) per low-res cell and (p, q, r) = coef[n][0..2][coff + c]
 * (the table of u3d_gn_bwd_finalize over Ctot channels; exact-2x levels use u3d_gn_bwd_apply with the constant 8 folded in). */
int u3d_gn_bwd_apply_children(int device, u3d_stream_t stream, const float* dlow, const float* x, const float* coef, int Ctot, int coff,
                              int N, int D1, int H1, int W1, int C, int ez, int ey, int ex, int relu_mask, float* out);
/* Per-(n, channel) sums of the nearest-upsampled image of x_low (N,D1,H1,W1,C) taken on the LOW-RES grid: stats[N][C][2] += (sum w x,
 * sum w x^2), w = children of a low-res voxel = (2 + [ez && z == 0]) (2 + [ey && y == 0]) (2 + [ex && x == 0]) — the GroupNorm statistics
 * of the upsampled half of such a level's virtual concat (u3d_gn_finalize takes them with scale 1). */
int u3d_chan_stats_children(int device, u3d_stream_t stream, const float* x_low, int N, int D1, int H1, int W1, int C, int ez, int ey,
                            int ex, double* stats);
int u3d_nearest_childsum_add(int device, u3d_stream_t stream, const float* dv, const float* x_low, float* dlow, double* gstats, int N,
                             int D, int H, int W, int D1, int H1, int W1, int C, const int32_t* zlo, const int32_t* ylo,
                             const int32_t* xlo, int cz, int cy, int cx);

/* The network's first convolution (in_channels 1..4 behind a one-group GroupNorm): K = 27*Cin is too small for
 * the MFMA tiling, so it has bandwidth-shaped kernels of its own (same semantics as u3d_conv3d / u3d_conv3d_wgrad;
 * x is a plain (N,D,H,W,Cin) tensor, w the reference (Cout,Cin,3,3,3) weights, Cin <= 4, Cout <= 32).
 *   _fwd : out = [relu](conv3d(x*a+b zero-padded, w)); out_stats (optional) as in u3d_conv3d
 *   _bwd : ONE pass over (dz, x) yields dw AND the GroupNorm-backward sums gstats[N][Cin][2] += (sum dg, sum dg*x)
 *          without computing the data gradient dg (see csrc/u3d_smallc.hip for the identity). */
int u3d_conv3d_small_cin_fwd(int device, u3d_stream_t stream, const float* x, const float* affine, const float* w,
                             float* out, int N, int D, int H, int W, int Cin, int Cout, int relu, double* out_stats);
/* ... with out_stats as `reps` replica rows [reps][N][Cout][2] (u3d_conv3d_ex_reps). */
int u3d_conv3d_small_cin_fwd_reps(int device, u3d_stream_t stream, const float* x, const float* affine, const float* w, float* out, int N,
                                  int D, int H, int W, int Cin, int Cout, int relu, double* out_stats, int reps);
size_t u3d_small_cin_bwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int u3d_conv3d_small_cin_bwd(int device, u3d_stream_t stream, const float* x, const float* affine, const float* dz,
                             const float* w, float* dw, double* gstats, int N, int D, int H, int W, int Cin, int Cout,
                             float* workspace, size_t workspace_floats);

/* Straightforward one-thread-per-output direct convolution (same semantics as u3d_conv3d with the
 * reference (Cout,Cin,27) weights, no packing).  Test/debug aid used to cross-check the MFMA kernel
 * on the device at sizes where the CPU oracle is slow.  flip=1 computes the data gradient. */
int u3d_conv3d_naive(int device, u3d_stream_t stream, const u3d_src_t* src, const float* w, float* out, int N,
                     int D, int H, int W, int Cin, int Cout, int relu, int flip);

/* ---- GroupNorm (nn.GroupNorm(G,C,eps=1e-5), buildingblocks.py:62-75) -------------------------
 * Statistics are per-(n,channel) sums accumulated in double (order-insensitive to < 1 ulp of fp32):
 *   u3d_chan_stats   stats[N][C][2] += (sum x, sum x^2) over all voxels of a (virtual) tensor
 *   u3d_gn_finalize  per (n,group) mean / biased var from up to two stat blocks (channels [0,C0) from
 *                    stats0 scaled by scale0, [C0,C0+C1) from stats1 scaled by scale1 — scale = 8 reuses
 *                    the low-res producer's sums for an exact 2x nearest upsampling), `count` voxels.
 *                    Writes affine[N][C][2] = (rstd*gamma, beta - mean*rstd*gamma) and mean_rstd[N][G][2].
 */
int u3d_chan_stats(int device, u3d_stream_t stream, const u3d_src_t* src, int N, int D, int H, int W,
                   double* stats);
/* ... into `reps` replica rows [reps][N][C][2] (zeroed by the caller): block b adds to row b % reps; the table is the sum of the rows
 * (u3d_gn_finalize_reps).  Same reason as u3d_conv3d_ex_reps: ~1000 blocks per sample adding to the same 2 C doubles. */
int u3d_chan_stats_reps(int device, u3d_stream_t stream, const u3d_src_t* src, int N, int D, int H, int W, double* stats, int reps);
int u3d_gn_finalize(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0,
                    const double* stats1, int C1, double scale1, int N, int G, double count, const float* gamma,
                    const float* beta, float eps, float* affine, float* mean_rstd);

/* GroupNorm backward, reduction part.  gstats[N][C][2] = (sum dg, sum dg*x) from u3d_conv3d.
 * Writes dgamma[C], dbeta[C] (summed over n) and the coefficient table coef[N][3][C] = (p,q,r) such that
 *   dx = p*dg + q*x + r        (the whole of GroupNorm backward is this per-(n,channel) affine map). */
int u3d_gn_bwd_finalize(int device, u3d_stream_t stream, const double* gstats, const float* mean_rstd,
                        const float* gamma, int N, int C, int G, double count, float* dgamma, float* dbeta,
                        float* coef);
/* The same two reductions for the first convolution of a decoder, whose input is the virtual concat cat(skip [0,C0), upsampled [C0,C0+C1))
 * (buildingblocks.py:491) and whose halves are read by DIFFERENT kernels as plain tensors (sub-pixel path, csrc/u3d_subpix.hip):
 *   u3d_gn_finalize_split      additionally writes compact tables affine_lo[N][Csplit][2] / affine_hi[N][C-Csplit][2] (either may be
 *                              NULL) — the rows of `affine` for the channels below / from Csplit (the skip / upsampled half; Csplit is
 *                              independent of how the statistics arrive) — instead of a strided-copy launch per half and direction;
 *   u3d_gn_bwd_finalize_split  takes the GroupNorm-backward sums as TWO tables gstats_lo[N][C0][2] / gstats_hi[N][C1][2] (the skip-half
 *                              and low-res data-gradient kernels each write their own) and additionally writes coef_hi[N][3][C1] =
 *                              (p, hi_scale*q, hi_scale*r) of the upper channels (NULL: none; hi_scale = 8: a low-res voxel of an exact
 *                              2x nearest upsampling stands for 8 children).  Needs N*C small enough for the kernel's LDS-staged path
 *                              (u3d_gn_bwd_finalize_split_supported == 1: every shipped configuration); identical arithmetic. */
int u3d_gn_finalize_split(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0, const double* stats1, int C1,
                          double scale1, int N, int G, double count, const float* gamma, const float* beta, float eps, float* affine,
                          float* mean_rstd, int Csplit, float* affine_lo, float* affine_hi);
/* u3d_gn_finalize_split on statistics tables with replica rows (u3d_conv3d_ex_reps): stats0 is [reps0][N][C0][2], stats1 [reps1][N][C1][2];
 * the rows are summed in ascending order before anything else (reps 1 = the plain table). */
int u3d_gn_finalize_reps(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0, int reps0, const double* stats1,
                         int C1, double scale1, int reps1, int N, int G, double count, const float* gamma, const float* beta, float eps,
                         float* affine, float* mean_rstd, int Csplit, float* affine_lo, float* affine_hi);
int u3d_gn_bwd_finalize_split_supported(int N, int C, int G);
int u3d_gn_bwd_finalize_split(int device, u3d_stream_t stream, const double* gstats_lo, int C0, const double* gstats_hi, int C1,
                              const float* mean_rstd, const float* gamma, int N, int G, double count, float* dgamma, float* dbeta,
                              float* coef, float hi_scale, float* coef_hi);

/* GroupNorm backward, elementwise part (+ fused ReLU backward of the producer, buildingblocks.py:47):
 *   out[n,v,c] = (p*dg[n,v,coff+c] + q*x[n,v,c] + r) * (relu_mask ? x>0 : 1),   c in [0,Cx)
 * dg has Cdg channels (Cdg > Cx for the skip half of a concat), x/out have Cx channels. */
int u3d_gn_bwd_apply(int device, u3d_stream_t stream, const float* dg, int Cdg, int coff, const float* x, int Cx,
                     const float* coef, int Ctot, int64_t voxels_per_n, int N, int relu_mask, float* out);

/* Same with an extra gradient term added BEFORE the mask: out = (p*dg + q*x + r + add) * mask.  ResNetBlock
 * (buildingblocks.py:277-288): the block input `residual` receives the GroupNorm-backward of conv2 plus the gradient
 * that flows through `out += residual`. */
int u3d_gn_bwd_apply_add(int device, u3d_stream_t stream, const float* dg, int Cdg, int coff, const float* x, int Cx,
                         const float* coef, int Ctot, int64_t voxels_per_n, int N, int relu_mask, const float* add,
                         float* out);

/* Same for the upsampled half of a concat: the backward of F.interpolate(nearest) (buildingblocks.py:614)
 * is a sum over the children of each low-res voxel, fused with the affine map and the ReLU mask of the
 * low-res producer:  out[n,v1,c] = (p*sum_children dg[.,coff+c] + cnt*(q*x1 + r)) * (x1>0).
 * zlo/ylo/xlo are device int32 tables of length D1+1/H1+1/W1+1: children of v1 are [lo[i], lo[i+1]). */
int u3d_gn_bwd_apply_up(int device, u3d_stream_t stream, const float* dg, int Cdg, int coff, const float* x1,
                        int C1, const float* coef, int Ctot, int N, int D, int H, int W, int D1, int H1, int W1,
                        const int32_t* zlo, const int32_t* ylo, const int32_t* xlo, int relu_mask, float* out);

/* ---- MaxPool3d(kernel_size=2) (buildingblocks.py:356): stride 2, floor ------------------------
 * fwd: out (N,D/2,H/2,W/2,C), argmax byte per output element (first max in z,y,x scan order, as ATen),
 *      optional out_stats as in u3d_conv3d.
 * bwd_merge: the gradient of the encoder feature e that feeds BOTH the pool and a skip connection:
 *      dz_e = (skip_grad + scatter(dpool)) * (e>0), with dpool = p*dg + q*pooled + r (GroupNorm backward
 *      of the pooled tensor fused, coef may be NULL => dpool = dg).  skip_grad may be NULL. */
int u3d_maxpool2_fwd(int device, u3d_stream_t stream, const float* x, int N, int D, int H, int W, int C,
                     float* out, uint8_t* argmax, double* out_stats);
int u3d_maxpool2_bwd_merge(int device, u3d_stream_t stream, const float* dg, const float* pooled,
                           const uint8_t* argmax, const float* coef, const float* skip_grad, const float* e,
                           int N, int D, int H, int W, int C, int relu_mask, float* out);
/* Same with the skip gradient computed on the fly as the GroupNorm backward of the decoder's first conv restricted to the
 * skip channels (the first C of Cdg channels of skip_dg, coefficient table skip_coef[N][3][Ctot]):
 *      skip_grad = ps*skip_dg[v, c] + qs*e[v, c] + rs
 * — the skip half of `torch.cat` backward + GroupNorm backward never makes a round trip through HBM. */
int u3d_maxpool2_bwd_merge_gn(int device, u3d_stream_t stream, const float* dg, const float* pooled,
                              const uint8_t* argmax, const float* coef, const float* skip_dg, int Cdg,
                              const float* skip_coef, int Ctot, const float* e, int N, int D, int H, int W, int C,
                              int relu_mask, float* out);

/* ---- final 1x1x1 conv with bias + Sigmoid/Softmax (model.py:88-101,141-147) -------------------
 * x (N,V,Cin) NDHWC; w (Cout,Cin), b (Cout).  logits/probs are written in the reference's NCDHW
 * layout (N,Cout,V).  act: 0 none, 1 sigmoid, 2 softmax over channels.  Cout <= 1024, Cin <= 256 (more than 16 outputs —
 * multi-class heads, model.py:88-91 allows any out_channels — run in tiles of 16 outputs; Softmax then takes a second pass over
 * the logits; round 5).
 * bwd: dlogits (N,Cout,V) -> dx (N,V,Cin) masked by (x>0) when relu_mask (x is a post-ReLU conv output);
 *      acc double[Cout*Cin + Cout] += (dw, db)  (convert with u3d_cvt_f64_f32). */
int u3d_conv1x1_head_fwd(int device, u3d_stream_t stream, const float* x, const float* w, const float* b, int N,
                         int64_t V, int Cin, int Cout, int act, float* logits, float* probs);
int u3d_conv1x1_head_bwd(int device, u3d_stream_t stream, const float* dlogits, const float* x, const float* w,
                         int N, int64_t V, int Cin, int Cout, int relu_mask, float* dx, double* acc);
/* ... with acc as `reps` replica rows [reps][Cout * Cin + Cout] (zeroed by the caller), folded by u3d_cvt_f64_f32_sum. */
int u3d_conv1x1_head_bwd_reps(int device, u3d_stream_t stream, const float* dlogits, const float* x, const float* w, int N, int64_t V,
                              int Cin, int Cout, int relu_mask, float* dx, double* acc, int reps);
int u3d_cvt_f64_f32(int device, u3d_stream_t stream, const double* src, float* dst, int64_t n);
/* dst[i] = float(src[0][i] + .. + src[reps - 1][i]), rows of n doubles, ascending order. */
int u3d_cvt_f64_f32_sum(int device, u3d_stream_t stream, const double* src, float* dst, int64_t n, int reps);

/* ---- optimizer step (trainer.py:246 `self.optimizer.step()`; create_optimizer, utils.py:246-316: torch.optim.Adam) ------------
 * The Adam update of ALL parameters in ONE launch (csrc/u3d_optim.hip): torch's multi-tensor form is 8 launches / 0.17 ms per step
 * for the 44 parameters of UNet3D f_maps=32.  descs: DEVICE array of n descriptors sorted by `first` = running element offset with
 * every parameter padded to a multiple of 4 elements (descriptor 0: 0); total_padded = the padded sum.  Per element, in torch's
 * operation order (torch/optim/adam.py, amsgrad=False, maximize=False, L2-style weight decay):
 *   g = grad + weight_decay*p;  m += (1-beta1)*(g-m);  v = beta2*v + (1-beta2)*g*g;  p -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps)
 * `step` is the 1-based step count of THIS update (all parameters of a launch share it). */
typedef struct {
    float* p;       /* parameter, updated in place */
    const float* g; /* its gradient */
    float* m;       /* exp_avg */
    float* v;       /* exp_avg_sq */
    int64_t first;
    int64_t numel;
} u3d_adam_desc_t;
int u3d_adam_step(int device, u3d_stream_t stream, const u3d_adam_desc_t* descs_device, int n, int64_t total_padded, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int64_t step);

/* ---- residual variants (ResidualUNet3D / ResidualUNetSE3D, model.py:193-278) -------------------------------------
 * 1x1x1 convolution WITH bias (ResNetBlock.conv1, buildingblocks.py:248-255).  x (N,V,Cin), w (Cout,Cin), y (N,V,Cout),
 * all NDHWC; out_stats as in u3d_conv3d (feeds conv2's GroupNorm).
 * bwd: dx (nullable) = dy * w; acc double[Cout*Cin + Cout] += (dw, db) — zeroed scratch, convert with u3d_cvt_f64_f32. */
int u3d_conv1x1_fwd(int device, u3d_stream_t stream, const float* x, const float* w, const float* bias, float* y, int N,
                    int64_t V, int Cin, int Cout, double* out_stats);
int u3d_conv1x1_bwd(int device, u3d_stream_t stream, const float* dy, const float* x, const float* w, int N, int64_t V,
                    int Cin, int Cout, float* dx, double* acc);
/* nn.ConvTranspose3d(Cin, Cout, kernel_size=3, stride=2, padding=1, bias=False) (buildingblocks.py:653-662):
 * x (N,D1,H1,W1,Cin) -> t (N,2D1-1,2H1-1,2W1-1,Cout); w in the reference layout (Cin,Cout,3,3,3).
 * bwd: dx (nullable) = stride-2 convolution of dt, masked by x > 0 when relu_mask; acc double[Cin*Cout*27] += dw.
 * packed / packed_t (optional, may be NULL): u3d_pack_convtr_weights images of w — mode 0 [tap][Cin][Cout] for the forward,
 * mode 1 [tap][Cout][Cin] for the data gradient (27*Cin*Cout floats each) — that make the weight-tile loads coalesced. */
int u3d_pack_convtr_weights(int device, u3d_stream_t stream, const float* w, int Cin, int Cout, int mode, float* packed);
int u3d_convtr3d_fwd(int device, u3d_stream_t stream, const float* x, const float* w, float* t, int N, int D1, int H1,
                     int W1, int Cin, int Cout, const float* packed);
/* The same forward on the sub-pixel MFMA kernel (csrc/u3d_subpix.hip): the 8 output parity classes (1/2/4/8 single taps each)
 * are accumulated from ONE staged input halo tile.  Cin % 4 == Cout % 4 == 0; packed = u3d_pack_convtr3d_subpixel image
 * (u3d_convtr3d_subpixel_packed_floats floats). */
long long u3d_convtr3d_subpixel_packed_floats(int Cin, int Cout);
int u3d_pack_convtr3d_subpixel(int device, u3d_stream_t stream, const float* w, int Cin, int Cout, float* packed);
int u3d_convtr3d_fwd_subpixel(int device, u3d_stream_t stream, const float* x, const float* packed, float* t, int N, int D1,
                              int H1, int W1, int Cin, int Cout);
int u3d_convtr3d_bwd(int device, u3d_stream_t stream, const float* dt, const float* x, const float* w, int N, int D1,
                     int H1, int W1, int Cin, int Cout, int relu_mask, float* dx, double* acc, const float* packed_t);
/* F.interpolate(t, size=skip.shape[2:]) (nearest, buildingblocks.py:650-651) + summation joining (:493):
 * out = skip + t[zmap[z], ymap[y], xmap[x]], out_stats as in u3d_conv3d.  bwd: dt[s] = sum of dj over the voxels mapped to
 * s (lo tables of length Dt+1 / Ht+1 / Wt+1: children of s are [lo[s], lo[s+1])); the skip's gradient is dj itself. */
int u3d_nearest_add_fwd(int device, u3d_stream_t stream, const float* skip, const float* t, const int32_t* zmap,
                        const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C,
                        float* out, double* out_stats);
int u3d_nearest_sum_bwd(int device, u3d_stream_t stream, const float* dj, const int32_t* zlo, const int32_t* ylo,
                        const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, float* dt);

/* ---- squeeze-and-excitation gates of ResNetBlockSE (se.py:18-114, buildingblocks.py:291-307) ------------------------
 * y: a block output (N,V,C) NDHWC, post-ReLU; C % 4 == 0, C <= 1024.  mode 0 scSE (max of both gates), 1 cSE, 2 sSE.
 *   gate_fwd : s[N][C] = channel means from conv3's fused statistics (ystats double[N][C][2], count = V),
 *              h[N][Cr] = relu(fc1 s + b1), gc[N][C] = sigmoid(fc2 h + b2)        (w1 (Cr,C), w2 (C,Cr) as nn.Linear)
 *   apply_fwd: a[N*V] = sigmoid(<ws, y[v,:]> + bs) (saved for backward), out = y * {max(gc,a) | gc | a}
 *   bwd_reduce: from dout (gradient of out): dls[N*V] = d logit_s, acc_gc double[N][C] += d gc, acc_ws double[C+1] +=
 *              (d ws, d bs) — zeroed scratch
 *   gate_bwd : FC backward -> ds[N][C] (gradient reaching y through the channel mean, already / V) and the parameter
 *              gradients dw1,db1,dw2,db2 (written); dz2 [N][C], dz1 [N][Cr] are scratch
 *   bwd_apply: m = (dout*gate + dls[v]*ws[c] + ds[n,c]) * (relu_mask ? y > 0 : 1) = gradient of the block's pre-ReLU sum */
int u3d_se_gate_fwd(int device, u3d_stream_t stream, const double* ystats, double count, const float* w1, const float* b1,
                    const float* w2, const float* b2, int N, int C, int Cr, float* s, float* h, float* gc);
int u3d_se_apply_fwd(int device, u3d_stream_t stream, const float* y, const float* gc, const float* ws, const float* bs, int N,
                     int64_t V, int C, int mode, float* out, float* a);
int u3d_se_bwd_reduce(int device, u3d_stream_t stream, const float* dout, const float* y, const float* gc, const float* a,
                      const float* ws, int N, int64_t V, int C, int mode, float* dls, double* acc_gc, double* acc_ws);
int u3d_se_gate_bwd(int device, u3d_stream_t stream, const double* acc_gc, const float* gc, const float* h, const float* s,
                    const float* w1, const float* w2, int N, int C, int Cr, double count, float* dz2, float* dz1, float* ds,
                    float* dw1, float* db1, float* dw2, float* db2);
int u3d_se_bwd_apply(int device, u3d_stream_t stream, const float* dout, const float* y, const float* gc, const float* a,
                     const float* ws, const float* dls, const float* ds, int N, int64_t V, int C, int mode, int relu_mask,
                     float* out);

/* ---- BCEDiceLoss / DiceLoss / BCEWithLogitsLoss on the logits (losses.py:187-201, :84-127, :11-37) ------------
 * loss = w_bce * mean(BCE-with-logits) + w_dice * (1 - mean_c dice_c),
 * dice_c = 2 * weight_c * sum(p*t) / clamp(sum(p^2) + sum(t^2), eps), p = sigmoid(logits), sums over (N, V) per channel.
 * BCEDiceLoss(alpha) = (w_bce 1, w_dice alpha); DiceLoss() = (0, 1); nn.BCEWithLogitsLoss() = (1, 0).
 * logits / target: (N, C, V) contiguous fp32 (the reference's NCDHW).  weight: optional device float[C] (DiceLoss's
 * per-class weight) or NULL.
 * fwd : sums double[u3d_bce_dice_scratch_doubles(N, C, V)] (scratch: per-block partial sums, need not be initialised; summed in a
 *       fixed order, so the loss is bit-reproducible), loss float[1], coef float[2C + 1] (saved for backward)
 * bwd : dlogits = grad_out[0] * dloss/dlogits; grad_out is a DEVICE scalar (NULL = 1) — no host synchronisation. */
long long u3d_bce_dice_scratch_doubles(int N, int C, int64_t V);
int u3d_bce_dice_fwd(int device, u3d_stream_t stream, const float* logits, const float* target, const float* weight,
                     int N, int C, int64_t V, float w_bce, float w_dice, float eps, double* sums, float* loss,
                     float* coef);
int u3d_bce_dice_bwd(int device, u3d_stream_t stream, const float* logits, const float* target, const float* coef,
                     const float* grad_out, int N, int C, int64_t V, float* dlogits);

/* ---- opt-in bf16-operand convolutions (BASELINE config 4: "bf16 compute", fp32 master weights) --------------
 * The same nn.Conv3d(in,out,3,padding=1,bias=False) (buildingblocks.py:56) and, with mode-1 packed weights on dy, its
 * data gradient — as u3d_conv3d, but on v_mfma_f32_32x32x16_bf16: operands rounded to bf16 (round-to-nearest-even; the
 * activations after the fp32 GroupNorm affine, while they are staged into LDS), products accumulated in FP32; inputs,
 * outputs, statistics and the residual are fp32 tensors.  Single-tensor sources only (no virtual concat); needs
 * Cin % 16 == 0 and Cout % 32 == 0 (u3d_conv3d_bf16_supported), 16-byte aligned pointers.
 *   x         (N,D,H,W,Cin) fp32        affine  optional (N,Cin,2) GroupNorm (a,b): conv input = a*x + b, zero padded
 *   packed_w  image written by u3d_pack_weights_bf16 (u3d_packed_weight_bf16_elems() 2-byte elements)
 *   out       (N,D,H,W,Cout) fp32 = [relu](conv + residual)
 *   out_stats optional double[N][Cout][2] += (sum, sum of squares) of the written values
 *   gx/gstats optional (data-gradient use): gx (N,D,H,W,Cout) fp32 = the forward input of the layer; gstats
 *             double[N][Cout][2] += (sum dg, sum dg*gx) — what GroupNorm backward needs.  Exclusive with out_stats. */
int u3d_conv3d_bf16_supported(int Cin, int Cout);
long long u3d_packed_weight_bf16_elems(int Cin, int Cout, int mode);
int u3d_pack_weights_bf16(int device, u3d_stream_t stream, const float* w, int Cout, int Cin, int mode, void* packed);
/* All bf16 images of a model in ONE launch, read and written at HBM rate (every optimizer step changes every weight: 36 + 36
 * images per config-4 step).  descs: DEVICE array of n u3d_pack_desc_t sorted by `first` = first BLOCK of the image within the
 * launch (descriptor 0: 0; an image takes u3d_pack_weights_bf16_blocks(Cin, Cout, mode) blocks), `packed` = the bf16 image
 * (u3d_packed_weight_bf16_elems elements), mode 0 / 1, cin_stride unused; total_blocks = the sum over the descriptors.
 * Modes 4 / 5 (round 5): the forward / data-gradient space-to-depth images of a ConvTranspose3d weight (Cin, Cout, 3,3,3) —
 * desc.Cin = Cin, desc.Cout = Cout, `packed` = u3d_convtr3d_t8_packed_elems(Cin, Cout, mode - 4) elements, bit for bit what
 * u3d_pack_convtr3d_t8 writes; needs Cin % 32 == 0 and Cout % 32 == 0 (u3d_pack_weights_bf16_blocks returns 0 otherwise).
 * Mode 6 (round 6): BOTH 3x3x3 images of one weight from ONE read of it (modes 0 and 1 each read the whole weight: a third of the
 * launch's bytes) — `packed` receives the mode-0 image and, u3d_packed_weight_bf16_elems(Cin, Cout, 0) elements behind it, the mode-1
 * image, bit for bit what the two modes write; needs Cin % 32 == 0 and Cout % 32 == 0 (blocks: (Cin / 32) * (Cout / 32) + 2). */
long long u3d_pack_weights_bf16_blocks(int Cin, int Cout, int mode);
int u3d_pack_weights_bf16_batch(int device, u3d_stream_t stream, const u3d_pack_desc_t* descs_device, int n, long long total_blocks);
int u3d_conv3d_bf16(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w, float* out,
                    int N, int D, int H, int W, int Cin, int Cout, int relu, double* out_stats, const float* gx,
                    double* gstats, const float* residual);
/* ... with an optional scratch buffer of u3d_conv3d_bf16_workspace_floats() floats (0 for shapes that never split): at the
 * bottom of the U (few tiles, many channels) the reduction over input channels is split over several blocks whose partial sums
 * are added in a fixed order by a second kernel that owns the epilogue — same results contract. */
long long u3d_conv3d_bf16_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
/* Host-only: which tile variant u3d_conv3d_bf16_ex (b16 = 0) / u3d_conv3d_bf16_ex_b16 (b16 = 1) runs for a shape —
 * (ksplit << 16) | (z-planes per tile: 4 or 8) << 8 | (32-channel n-tiles per block: 1 or 2) << 4 | blocks per CU (2 or 3);
 * planes = 5 (b16 only, round 5): the FLAT 5 x 10 x 10 tile — 500 voxels in raster order are the GEMM rows of one block, one block per
 * CU, always with ksplit > 1 — which the small wide levels take when it executes at most 3/4 of the padded rows of the 4 x 8 x 8 tiling
 * (config 4's 10 x 20 x 20 and 5 x 10 x 10 levels; u3d_set_tuning key 16 = 1: never);
 * -1 for unsupported channel counts.  No reference counterpart (ATen picks its MIOpen / oneDNN algorithm internally, behind
 * buildingblocks.py:56); the parity tests use it to assert that a pinned shape runs the variants the benchmark shape runs. */
int u3d_conv3d_bf16_tile_variant(int N, int D, int H, int W, int Cin, int Cout, int b16);
int u3d_conv3d_bf16_ex(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w, float* out,
                       int N, int D, int H, int W, int Cin, int Cout, int relu, double* out_stats, const float* gx,
                       double* gstats, const float* residual, float* workspace, long long workspace_floats);

/* ---- opt-in "split fp32" convolutions: FP32 operands on the bf16 matrix pipe --------------------------------------------
 * The same nn.Conv3d / data gradient as u3d_conv3d with FP32-grade results: every operand is split exactly into three bf16
 * values (a = a_h + a_m + a_l) and the six partial products down to 2^-16 of a*b are accumulated in FP32 on
 * v_mfma_f32_32x32x16_bf16 (16x the rate of the fp32 MFMA: 6/16 of its pipe time); the dropped terms are ~2^-23 relative, the
 * size of one rounding of an fp32 accumulation (csrc/u3d_bf16.hip; measured against float64 next to u3d_conv3d in
 * tests/test_gpu_f32s.py).  Arguments as u3d_conv3d_bf16_ex; packed_w from u3d_pack_weights_f32s
 * (u3d_packed_weight_f32s_elems() 2-byte elements: three images), which can read a channel slice [ci_off, ci_off + Cin) of
 * rows that are `ld` input channels wide (the skip half of a decoder's first convolution).  Workspace:
 * u3d_conv3d_bf16_workspace_floats(). */
long long u3d_packed_weight_f32s_elems(int Cin, int Cout, int mode);
int u3d_pack_weights_f32s(int device, u3d_stream_t stream, const float* w, int Cout, int Cin, int mode, int ld, int ci_off,
                          void* packed);
int u3d_conv3d_f32s(int device, u3d_stream_t stream, const float* x, const float* affine, const void* packed_w, float* out, int N,
                    int D, int H, int W, int Cin, int Cout, int relu, double* out_stats, const float* gx, double* gstats,
                    const float* residual, float* workspace, long long workspace_floats);

/* Weight gradient of the same convolution with bf16 operands / FP32 accumulation (autograd of trainer.py:245 for
 * buildingblocks.py:56): dw (Cout,Cin,3,3,3) fp32, reference layout, = sum over voxels of g(x)[v+tap] (x) dz[v] with
 * g = a*x + b zero padded.  Needs Cin % 32 == 0, Cout % 32 == 0 (Cout % 64 == 32: 64-column blocks whose upper half is read as zero)
 * and a scratch buffer of
 * u3d_wgrad_bf16_workspace_floats() floats (split partial sums, reduced in a fixed order: run-to-run identical). */
int u3d_conv3d_wgrad_bf16_supported(int Cin, int Cout);
long long u3d_wgrad_bf16_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int u3d_conv3d_wgrad_bf16(int device, u3d_stream_t stream, const float* x, const float* affine, const float* dz, float* dw,
                          int N, int D, int H, int W, int Cin, int Cout, float* workspace, long long workspace_floats);

/* nn.ConvTranspose3d(in, out, 3, stride=2, padding=1, bias=False) (buildingblocks.py:653-662) and its two gradients on the bf16
 * MFMA kernels, in SPACE-TO-DEPTH form: the (2n-1)^3 output is stored as T8 (N,D1,H1,W1, 8*Cout) with
 * T8[i][p*Cout + co] = t[2i + p][co] (p = pz*4 + py*2 + px), which turns all three passes into 2x2x2 convolutions on the low-res
 * grid (64/27 = 2.4x the minimal multiply-adds).  u3d_nearest_add_fwd_t8 / u3d_nearest_sum_bwd_t8 are u3d_nearest_add_fwd /
 * _sum_bwd reading / writing that layout (T8 entries outside the (2n-1) grid are written as zero, never read).
 * Needs Cin % 32 == 0 and Cout % 8 == 0.  packed: u3d_pack_convtr3d_t8 image of w (Cin,Cout,3,3,3), mode 0 forward, 1 data
 * gradient.  x_mask (optional): dx = x_mask > 0 ? dx : 0 (ReLU mask of the tensor that was upsampled). */
int u3d_convtr3d_t8_supported(int Cin, int Cout);
long long u3d_convtr3d_t8_packed_elems(int Cin, int Cout, int mode);
int u3d_pack_convtr3d_t8(int device, u3d_stream_t stream, const float* w, int Cin, int Cout, int mode, void* packed);
int u3d_convtr3d_fwd_t8(int device, u3d_stream_t stream, const float* x, const void* packed, float* t8, int N, int D1, int H1,
                        int W1, int Cin, int Cout);
int u3d_convtr3d_dgrad_t8(int device, u3d_stream_t stream, const float* dt8, const void* packed, const float* x_mask, float* dx,
                          int N, int D1, int H1, int W1, int Cin, int Cout);
long long u3d_convtr3d_wgrad_t8_workspace_floats(int N, int D1, int H1, int W1, int Cin, int Cout);
int u3d_convtr3d_wgrad_t8(int device, u3d_stream_t stream, const float* x, const float* dt8, float* dw, int N, int D1, int H1,
                          int W1, int Cin, int Cout, float* workspace, long long workspace_floats);
int u3d_nearest_add_fwd_t8(int device, u3d_stream_t stream, const float* skip, const float* t8, const int32_t* zmap,
                           const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C,
                           float* out, double* out_stats);
int u3d_nearest_sum_bwd_t8(int device, u3d_stream_t stream, const float* dj, const int32_t* zlo, const int32_t* ylo,
                           const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, float* dt8);

/* ---- layer orders other than 'gcr' (buildingblocks.py:10-96): LeakyReLU / ELU, GroupNorm after the convolution --------
 * Activation codes: 0 none, 1 nn.ReLU, 2 nn.LeakyReLU(slope) (buildingblocks.py:49: 0.01; ResNetBlock :271: 0.1), 3 nn.ELU
 * (alpha 1, :51).  The convolutions run with their epilogue ReLU off; these passes supply the rest:
 *   u3d_act_fwd         out = f(x), n elements (in place allowed)
 *   u3d_act_bwd         out = g * f'(.) with the derivative expressed through the OUTPUT y of f (in place on g allowed)
 *   u3d_affine_act_fwd  out[n,v,c] = f(a[n,c]*z[n,v,c] + b[n,c]): nn.GroupNorm apply (+ non-linearity) of a post-norm layer
 *                       ('cgr' family, GroupNorm on the conv OUTPUT channels, buildingblocks.py:62-66); affine (N,C,2)
 *   u3d_affine_add_act_fwd  out = f(a*z + b + add[n,v,c]) (add may be NULL): the tail of ResNetBlock.forward for post-norm
 *                       orders — conv3's GroupNorm, `out += residual`, non-linearity (buildingblocks.py:277-288)
 *   u3d_pair_stats      stats double[N][C][2] += (sum_v a, sum_v a*b): the reductions nn.GroupNorm's backward needs */
int u3d_act_fwd(int device, u3d_stream_t stream, const float* x, int64_t n, int mode, float slope, float* out);
int u3d_act_bwd(int device, u3d_stream_t stream, const float* g, const float* y, int64_t n, int mode, float slope, float* out);
int u3d_affine_act_fwd(int device, u3d_stream_t stream, const float* z, const float* affine, int N, int64_t V, int C, int mode,
                       float slope, float* out);
int u3d_affine_add_act_fwd(int device, u3d_stream_t stream, const float* z, const float* affine, const float* add, int N,
                           int64_t V, int C, int mode, float slope, float* out);
int u3d_pair_stats(int device, u3d_stream_t stream, const float* a, const float* b, int N, int64_t V, int C, double* stats);

/* ---- F.interpolate(mode='trilinear' | 'area') to the skip's size (InterpolateUpsampling, buildingblocks.py:598-614; ATen
 * upsample_trilinear3d align_corners=False / adaptive_avg_pool3d): separable, <= 2 source samples per output index and
 * dimension.  Per-dimension host tables (engine.resample_tables_host, ATen's float32 formulas): idx int32[2*n_out] source
 * indices, wt float[2*n_out] weights, rng int32[2*n_in] = [lo,hi) outputs touching each input.  x (N,D1,H1,W1,C) ->
 * out (N,D,H,W,C); bwd = the adjoint as a fixed-order gather (no atomics): dx (N,D1,H1,W1,C) from dout (N,D,H,W,C). */
int u3d_resample2_fwd(int device, u3d_stream_t stream, const float* x, const int32_t* idx_z, const int32_t* idx_y,
                      const int32_t* idx_x, const float* wt_z, const float* wt_y, const float* wt_x, int N, int D1, int H1, int W1,
                      int D, int H, int W, int C, float* out);
int u3d_resample2_bwd(int device, u3d_stream_t stream, const float* dout, const int32_t* rng_z, const int32_t* rng_y,
                      const int32_t* rng_x, const int32_t* idx_z, const int32_t* idx_y, const int32_t* idx_x, const float* wt_z,
                      const float* wt_y, const float* wt_x, int N, int D1, int H1, int W1, int D, int H, int W, int C, float* dx);

/* ---- concat joining for residual decoders with an explicit upsample='deconv' (buildingblocks.py:435-468, :491): the joined
 * tensor out (N,D,H,W,Cs+Ct) = cat(skip, nearest-resized t) feeds the block's 1x1x1 convolution; u3d_split_channels is its
 * backward routing (skip gradient | gradient of the resized tensor). */
int u3d_nearest_cat_fwd(int device, u3d_stream_t stream, const float* skip, const float* t, const int32_t* zmap, const int32_t* ymap,
                        const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt, int Cs, int Ct, float* out);
int u3d_split_channels(int device, u3d_stream_t stream, const float* x, int64_t rows, int C0, int C1, float* out0, float* out1);

/* ---- the remaining layer-order characters (create_conv, buildingblocks.py:10-96): 'b' nn.BatchNorm3d (:78-88), the conv bias of
 * layers without a norm (:54-55: 'cr', 'cl', 'ce', 'c'), 'd' / 'D' dropout (:89-92).
 *   u3d_bn_finalize      per-(n,c) sums (as u3d_gn_finalize: two channel ranges with scales) -> per-CHANNEL batch mean / biased
 *                        variance (training) or the running statistics (eval), the (a, b) table affine[N][C][2] the convolutions
 *                        apply, mean_rstd[C][2]; training also updates running_mean / running_var in place like ATen
 *                        (momentum, unbiased variance); count = voxels per sample
 *   u3d_bn_bwd_finalize  gstats[N][C][2] = (sum dg, sum dg*x) -> dgamma, dbeta, coef[N][3][C] for u3d_gn_bwd_apply*
 *   u3d_bias_table       affine[N][C][2] = (1, bias[c]): a layer without a norm is "post-norm with a constant affine"
 *   u3d_bias_grad        dbias[c] = sum_n stats[n][c][0] (stats from u3d_pair_stats on the pre-activation gradient)
 *   u3d_mul              out = a * b elementwise (dropout mask, drawn by the caller with torch's generator) */
int u3d_bn_finalize(int device, u3d_stream_t stream, const double* stats0, int C0, double scale0, const double* stats1, int C1,
                    double scale1, int N, double count, const float* gamma, const float* beta, float eps, int training,
                    float momentum, float* running_mean, float* running_var, float* affine, float* mean_rstd);
int u3d_bn_bwd_finalize(int device, u3d_stream_t stream, const double* gstats, const float* mean_rstd, const float* gamma, int N,
                        int C, double count, int training, float* dgamma, float* dbeta, float* coef);
int u3d_bias_table(int device, u3d_stream_t stream, const float* bias, int N, int C, float* affine);
int u3d_bias_grad(int device, u3d_stream_t stream, const double* stats, int N, int C, float* dbias);
int u3d_mul(int device, u3d_stream_t stream, const float* a, const float* b, int64_t n, float* out);

/* ---- bf16 ACTIVATION STORAGE (`activation_dtype: bf16`, BASELINE config 4: ResidualUNet3D with bf16 compute) ------------------
 * The same operators with every NDHWC activation / gradient tensor stored as bf16 (`void*` arguments below; what
 * torch.autocast(bfloat16) leaves in memory between buildingblocks.py:277-288's operators): half the HBM bytes of every
 * bandwidth-bound pass and of the convolutions' operand traffic.  Unchanged: fp32 arithmetic and accumulation, fp32 parameters,
 * weight gradients and affine / coefficient tables, f64 statistics (taken over the values AS STORED, so the consuming GroupNorm
 * normalises exactly the tensor it reads), fp32 logits / probabilities, fp32 split-K scratch.  Stores round to nearest even.
 * Argument meaning = the entry point of the same name without the suffix.  Channel counts: multiples of 4 (8-byte quads). */
int u3d_conv3d_bf16_ex_b16(int device, u3d_stream_t stream, const void* x, const float* affine, const void* packed_w, void* out,
                           int N, int D, int H, int W, int C, int K, int relu, double* out_stats, const void* gx, double* gstats,
                           const void* residual, float* workspace, long long workspace_floats);
int u3d_conv3d_wgrad_bf16_b16(int device, u3d_stream_t stream, const void* x, const float* affine, const void* dz, float* dw, int N,
                              int D, int H, int W, int C, int K, float* workspace, long long workspace_floats);
/* Host-only: which kernel u3d_conv3d_wgrad_bf16_b16 runs for a shape — 16: 2 x 8 x 16-voxel tiles (bit-identical to
 * u3d_conv3d_wgrad_bf16 fed with the same bf16-representable values), 8: 4 x 8 x 8-voxel tiles (fewer wasted voxels on extents
 * that are not multiples of 16; another summation order over the voxels), 0: the round-3 kernel (tensors beyond 2 GiB), -1:
 * unsupported channel counts.  No reference counterpart (ATen picks its algorithm behind buildingblocks.py:56). */
int u3d_conv3d_wgrad_bf16_b16_variant(int N, int D, int H, int W, int C, int K);
/* u3d_conv3d_wgrad_bf16 / _b16 with the GroupNorm-backward reduction of the layer's INPUT (u3d_gn_bwd_job_t, see u3d_conv3d_wgrad_job:
 * what u3d_gn_bwd_finalize / _split would do in a launch of their own) as one extra block of the launch that adds the weight gradient's
 * splits — config 4 ran 18 single-block finalize launches of ~7 us per step behind its weight gradients (buildingblocks.py:75 under
 * autograd).  Same bits as the separate launch.  Only a shape whose weight gradient HAS a reduce launch can carry the job (with one split
 * — config 4's 1024-channel level — the main kernel writes dw itself): u3d_conv3d_wgrad_bf16_job_supported(shape, b16 = 0 / 1 for the
 * fp32- / bf16-storage entry point, the dw pointer, the job's N, C0 + C1, G) says so (host-only); job = NULL is the plain entry point. */
int u3d_conv3d_wgrad_bf16_job_supported(int N, int D, int H, int W, int C, int K, int b16, const float* dw, int jobN, int jobC, int jobG);
int u3d_conv3d_wgrad_bf16_job(int device, u3d_stream_t stream, const float* x, const float* affine, const float* dz, float* dw, int N,
                              int D, int H, int W, int Cin, int Cout, float* workspace, long long workspace_floats,
                              const u3d_gn_bwd_job_t* job);
int u3d_conv3d_wgrad_bf16_b16_job(int device, u3d_stream_t stream, const void* x, const float* affine, const void* dz, float* dw, int N,
                                  int D, int H, int W, int C, int K, float* workspace, long long workspace_floats,
                                  const u3d_gn_bwd_job_t* job);
int u3d_convtr3d_fwd_t8_b16(int device, u3d_stream_t stream, const void* x, const void* packed, void* t8, int N, int D1, int H1,
                            int W1, int Cl, int Cs);
int u3d_convtr3d_dgrad_t8_b16(int device, u3d_stream_t stream, const void* dt8, const void* packed, const void* x_mask, void* dx,
                              int N, int D1, int H1, int W1, int Cl, int Cs);
/* Forward with scratch (round 5): on small low-res grids with many channels (config 4's 1024 -> 512 level on 5 x 10 x 10) the flat
 * 5 x 10 x 10 tile of the bf16-storage convolutions runs the 2x2x2 kernel with a split channel reduction — one tile instead of eight
 * 3/4-padded ones.  u3d_convtr3d_fwd_t8_workspace_floats() = 0 where the plan of u3d_convtr3d_fwd_t8_b16 stays; NULL / too small = that plan.
 * Same contract (buildingblocks.py:617-664: ConvTranspose3d k3 s2 p1 before the resize + join). */
long long u3d_convtr3d_fwd_t8_workspace_floats(int N, int D1, int H1, int W1, int Cl, int Cs);
int u3d_convtr3d_fwd_t8_b16_ex(int device, u3d_stream_t stream, const void* x, const void* packed, void* t8, int N, int D1, int H1,
                               int W1, int Cl, int Cs, float* workspace, long long workspace_floats);
/* Split-K forms of the two data-gradient entry points above (the contraction runs over 8*Cs channels: hundreds of serial 16-channel
 * chunks on few blocks at the bottom of the U).  `workspace`: u3d_convtr3d_dgrad_t8_workspace_floats() fp32 elements, or NULL / too small
 * = unsplit.  Partial sums are added in a fixed order by the reduction kernel of u3d_conv3d_bf16_ex, which applies the ReLU mask. */
long long u3d_convtr3d_dgrad_t8_workspace_floats(int N, int D1, int H1, int W1, int Cl, int Cs);
int u3d_convtr3d_dgrad_t8_ex(int device, u3d_stream_t stream, const float* dt8, const void* packed, const float* x_mask, float* dx,
                             int N, int D1, int H1, int W1, int Cl, int Cs, float* workspace, long long workspace_floats);
int u3d_convtr3d_dgrad_t8_b16_ex(int device, u3d_stream_t stream, const void* dt8, const void* packed, const void* x_mask, void* dx,
                                 int N, int D1, int H1, int W1, int Cl, int Cs, float* workspace, long long workspace_floats);
int u3d_convtr3d_wgrad_t8_b16(int device, u3d_stream_t stream, const void* x, const void* dt8, float* dw, int N, int D1, int H1,
                              int W1, int Cl, int Cs, float* workspace, long long workspace_floats);
/* x_is_f32: the block input is the fp32 network input (first encoder block) */
int u3d_conv1x1_fwd_b16(int device, u3d_stream_t stream, const void* x, int x_is_f32, const float* w, const float* bias, void* y,
                        int N, int64_t V, int Cin, int Cout, double* out_stats);
int u3d_conv1x1_bwd_b16(int device, u3d_stream_t stream, const void* dy, const void* x, int x_is_f32, const float* w, int N,
                        int64_t V, int Cin, int Cout, void* dx, double* acc);
/* ResNetBlock.conv1 (1x1x1, bias; reference buildingblocks.py:248-255) under bf16 storage on v_mfma_f32_32x32x16_bf16: the
 * weights are rounded to bf16 like every MFMA operand of the bf16 path; x, y, dy, dx are bf16 (N, V, C) tensors.
 * _supported: Cin, Cout powers of two in 64..1024.  out_stats (N, Cout, 2) += (sum, sum of squares) of the STORED y.
 * _bwd: dx (may be NULL) = dy W; dw (Cout, Cin) and db (Cout) are WRITTEN (fp32, fixed-order sums: run-to-run identical). */
int u3d_conv1x1_mfma_b16_supported(int Cin, int Cout);
int u3d_conv1x1_fwd_mfma_b16(int device, u3d_stream_t stream, const void* x, const float* w, const float* bias, void* y, int N,
                             int64_t V, int Cin, int Cout, double* out_stats);
long long u3d_conv1x1_bwd_mfma_b16_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int u3d_conv1x1_bwd_mfma_b16(int device, u3d_stream_t stream, const void* dy, const void* x, const float* w, int N, int D, int H,
                             int W, int Cin, int Cout, void* dx, float* dw, float* db, float* workspace,
                             long long workspace_floats);
int u3d_maxpool2_fwd_b16(int device, u3d_stream_t stream, const void* x, int N, int D, int H, int W, int C, void* out,
                         uint8_t* argmax);
int u3d_maxpool2_bwd_merge_b16(int device, u3d_stream_t stream, const void* dg, const void* pooled, const uint8_t* argmax,
                               const float* coef, const void* skip_grad, const void* e, int N, int D, int H, int W, int C,
                               int relu_mask, void* out);
int u3d_nearest_add_fwd_t8_b16(int device, u3d_stream_t stream, const void* skip, const void* t8, const int32_t* zmap,
                               const int32_t* ymap, const int32_t* xmap, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C,
                               void* out, double* out_stats);
int u3d_nearest_sum_bwd_t8_b16(int device, u3d_stream_t stream, const void* dj, const int32_t* zlo, const int32_t* ylo,
                               const int32_t* xlo, int N, int D, int H, int W, int Dt, int Ht, int Wt, int C, void* dt8);
/* add may be NULL (u3d_gn_bwd_apply) */
int u3d_gn_bwd_apply_b16(int device, u3d_stream_t stream, const void* dg, int Cdg, int coff, const void* x, int Cx, const float* coef,
                         int Ctot, int64_t voxels_per_n, int N, int relu_mask, const void* add, void* out);
int u3d_conv1x1_head_fwd_b16(int device, u3d_stream_t stream, const void* x, const float* w, const float* b, int N, int64_t V,
                             int Cin, int Cout, int act, float* logits, float* probs);
int u3d_conv1x1_head_bwd_b16(int device, u3d_stream_t stream, const float* dlogits, const void* x, const float* w, int N, int64_t V,
                             int Cin, int Cout, int relu_mask, void* dx, double* acc);
/* squeeze-and-excitation gates (se.py:18-114, ResNetBlockSE buildingblocks.py:291-307) on bf16 block outputs: y / out / dout / the
 * returned gradient are bf16; the channel gate gc, the spatial gate a, d logit_s and every sum stay fp32 / f64 (u3d_se_gate_fwd and
 * u3d_se_gate_bwd touch no activation tensor and serve both storage modes) */
int u3d_se_apply_fwd_b16(int device, u3d_stream_t stream, const void* y, const float* gc, const float* ws, const float* bs, int N,
                         int64_t V, int C, int mode, void* out, float* a);
int u3d_se_bwd_reduce_b16(int device, u3d_stream_t stream, const void* dout, const void* y, const float* gc, const float* a,
                          const float* ws, int N, int64_t V, int C, int mode, float* dls, double* acc_gc, double* acc_ws);
int u3d_se_bwd_apply_b16(int device, u3d_stream_t stream, const void* dout, const void* y, const float* gc, const float* a,
                         const float* ws, const float* dls, const float* ds, int N, int64_t V, int C, int mode, int relu_mask,
                         void* out);

/* ---- layout: NCDHW <-> NDHWC for multi-channel model inputs ------------------------------------ */
int u3d_ncdhw_to_ndhwc(int device, u3d_stream_t stream, const float* src, float* dst, int N, int C, int64_t V);
int u3d_ndhwc_to_ncdhw(int device, u3d_stream_t stream, const float* src, float* dst, int N, int C, int64_t V);

#ifdef __cplusplus
}
#endif
#endif /* U3D_H */
