// Shared helpers for libpcgc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pcgc.h"

namespace pcgc {

void set_error(const char* fmt, ...);

#define PCGC_CHECK_HIP(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      ::pcgc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                        __FILE__, __LINE__);                                        \
      return -100;                                                                  \
    }                                                                               \
  } while (0)

#define PCGC_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::pcgc::set_error(__VA_ARGS__);      \
      return -1;                           \
    }                                      \
  } while (0)

inline int launch_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return -101;
  }
  return 0;
}

// Description of one convolution launch (all strides in floats).
struct ConvArgs {
  const float* x;      // [B, Din^3, x_cs], channels [x_co, x_co+Cin) are read
  const float* w;      // TF layout for the direct kernel; packed layout for MFMA kernels
  const float* bias;   // [Cout] or nullptr
  float* y;            // [B, Dout^3, y_cs], channels [y_co, y_co+Cout) are written
  const float* res;    // optional residual, same geometry as y (y_cs, y_co); out = relu(res + act(conv))
  int B, Din, Dout, Cin, Cout;
  int x_cs, x_co, y_cs, y_co;
  int ksize;           // 1 or 3
  int mode;            // 0 stride-1, 1 stride-2 conv (pad 0/1), 2 stride-2 transposed conv
  int relu;            // ReLU on conv+bias
  int absval;          // |.| on conv+bias (hyper decoder scale head), applied before lower bound
  float lower_bound;   // max(., lower_bound) when absval
  // second, fused 1x1x1 convolution of a VRN block (vrn_mfma.hip); unused elsewhere
  const float* w2;     // fuse 1: conv2_1 packed MFMA weights (Cin -> C/4 on the same input tile)
                       // fuse 2: conv2_3 TF weights [C/4][C/2] applied to relu(conv2_2 + bias)
  const float* bias2;
  float* y2;           // fuse 1: tensor2_1 output [B, D^3, y2_cs]
  int y2_cs, cout2;
  // "Q4" layout [b][d][h][C/4][w][4] (vrn_row.hip) instead of NDHWC for the input / the output (+ residual); only the
  // kernels of conv_mfma.hip read these flags, every other launcher refuses a Q4 tensor
  int x_q4 = 0, y_q4 = 0;
  // bwd-data epilogue of the training step: out = (mask > 0) ? (add_to + conv) : 0.  mask = the forward layer's input
  // (a ReLU output: its gradient passes only where it is positive), add_to = gradient already accumulated for that
  // tensor (may alias y).  Both laid out like y (y_cs, y_co).
  const float* mask = nullptr;
  const float* add_to = nullptr;
};

int launch_conv_direct(const ConvArgs& a, hipStream_t s);
// returns 1 if an MFMA kernel exists for this shape (and was launched when run=true), 0 if not, <0 error
int launch_conv_mfma(const ConvArgs& a, const float* packed_w, hipStream_t s, bool run);
// two independent stride-1 layers of the small-launch family in one launch: 1 launched, 0 no pair kernel for these shapes, < 0 error
int launch_conv_mfma_pair(const ConvArgs& a, const float* packed_a, const ConvArgs& b, const float* packed_b, hipStream_t s);
// layer a, then the 1x1x1 layer b accumulating into a's output (b.add_to == b.y == a.y): the same return codes
int launch_conv_mfma_chain(const ConvArgs& a, const float* packed_a, const ConvArgs& b, const float* packed_b, hipStream_t s);
// a 1x1x1 layer writing t23 (C / 2 channels) and the block's merge out = relu(x + [t12 | t23]) on the same tiles: the same return codes
struct MergeArgs {
  const float* x;      // block input [B, D^3, C]
  const float* t12;    // first path's end [B, D^3, C / 2]
  float* out;          // [B, D^3, C]
  int C;
};
int launch_conv_mfma_merge(const ConvArgs& a, const float* packed_a, const MergeArgs& m, hipStream_t s);
// tap-split (one filter slice per wave) row-packed kernels with optional VRN fusions, vrn_mfma.hip.
// fuse: 0 plain (+ residual epilogue), 1 also emits tensor2_1 = relu(conv2_1(x)), 2 applies conv2_3 + residual
// to the result.  Same packed-weight layout as launch_conv_mfma.  Returns 1 launched, 0 unsupported, <0 error.
int launch_conv_ks(const ConvArgs& a, const float* packed_w, int fuse, hipStream_t s, bool run);
// LDS-tiled VALU direct conv (conv_valu.hip): a.w = TF-layout weights.  1 launched, 0 unsupported, <0 error.
int launch_conv_valu(const ConvArgs& a, hipStream_t s, bool run);
// C = 16 Voxception-ResNet block as two VALU kernels (vrn_valu.hip).  w = {w11,b11,w12,b12,w21,b21,w22,b22,w23,b23}
// in TF layouts; which 0 = [conv1_1|conv2_1] -> t12, 1 = [conv1_2 | conv2_2+conv2_3] + residual -> out.
int launch_vrn16_valu(const float* x, float* t12, float* out, const float* const* w, int B, int D, int which, hipStream_t s);
// Row kernels on v_mfma_f32_4x4x1_16B_f32 for the 64^3 / C = 16 stage (vrn_row.hip); tensors in the Q4 layout.
// launch_vrn16_row: which 0 = [conv1_1|conv2_1] -> t12, 1 = [conv1_2 | conv2_2+conv2_3] + residual -> out (may alias x).
// x_nonneg: the caller vouches that x >= 0 everywhere (the block follows a ReLU layer): BC drops its final ReLU.
// Exact skipping of empty space (analysis, 64^3 stage).  Wherever the receptive field of a wave tile holds no occupied
// voxel, the tile equals — bit for bit: same inputs in the same positions, same kernel, same summation order, cube faces
// included — the same tile of the kernel's output for an EMPTY cube, and the wave copies that instead of computing it.
// A launch keeps its geometry (one wave per tile); which tile a wave takes comes from `order`, a permutation of the tile
// indices with the tiles that must be computed FIRST (in their natural order) and the empty ones after them: workgroups
// are dispatched in index order, so the heavy waves spread evenly over the SIMDs and the copy waves fill in behind them
// (with the natural order a SIMD that happened to hold two heavy waves set the launch's duration: no gain at all).
struct RowSkip {
  const unsigned* order = nullptr;              // [tiles of the launch]: wave i takes tile order[i]; nullptr: no skipping
  const unsigned* n_heavy = nullptr;            // device word: waves >= *n_heavy copy their tile from `empty`
  const float* empty = nullptr;                 // this kernel's output for an all-zero cube, laid out like one cube of it
  unsigned* counter = nullptr;                  // optional (tests): += 1 per skipped wave tile
  // materialize = 1: an empty tile is copied from `empty`, every tensor stays complete.  0: an empty tile is NOT WRITTEN
  // (a "virtual" tile: neither its bytes nor its producer's time exist); whoever reads that tensor must be told, per
  // row, to read the producer's empty-cube response instead.  2: copied unless the tile order marks it kTileUnread (its
  // reader skips every tile that would touch it: TileCfg::need):
  int materialize = 1;
  // in_*: the tensor this kernel reads WITH its halo (kernel A: the block input; kernel BC: tensor1_1 | tensor2_1);
  // res_*: kernel BC's residual input.  virtual[b * 64 + d] bit h = row (d, h) lies in a tile its producer did not write;
  // *_empty = that producer's empty-cube response.  nullptr: the tensor is complete.
  const unsigned long long* in_virtual = nullptr;
  const float* in_empty = nullptr;
  const unsigned long long* res_virtual = nullptr;
  const float* res_empty = nullptr;
};
// One launch's tile geometry and the receptive field of its output, for launch_tile_order.  A tile covers th rows x ld
// planes of a (64 / step)^3 grid; output row / plane o depends on the fine (64^3) rows / planes [step * o - lo,
// step * o + hi] of the cube: step 1, lo = hi = radius for the 64^3 stage; step 2 behind down_1 (pads 0 in front and
// 1 behind: output o reads fine 2o .. 2o + 2 of a tensor whose own radius is 7, then every 32^3 layer adds 2 fine voxels)
struct TileCfg {
  int th, ld, lo, hi, step;
  // need > 0 (the stage's last launch, whose empty tiles are copied for a reader that knows nothing of virtual rows): an
  // empty tile whose window dilated to `need` instead of lo / hi holds no occupied row either is read by NO tile its reader
  // computes — its entry in the tile order carries kTileUnread and the launch leaves it alone (RowSkip::materialize = 2)
  int need = 0;
};
constexpr unsigned kTileUnread = 0x80000000u;
constexpr int kSkipLaunches = 8;                // conv_in, A / BC of the three C = 16 blocks, down_1
constexpr int kSkipLaunchesMid = 6;             // A / BC of the three C = 32 blocks (32^3 stage of the analysis)
// Tile orders of every chunk of `chunk` cubes among `total` cubes for the n_cfg launch configurations of a stage, in one
// launch: chunk k (first cube c0 = k * chunk, n cubes) gets order + (c0 * n_cfg + cfg * n) * tiles_cap, n_heavy[k * n_cfg + cfg]
// and — 64^3 stage — the table virt + (c0 * n_cfg + cfg * n) * 64 of rows that lie in its empty tiles (RowSkip::in_virtual).
// cfg_small (optional): the configurations for chunks of <= 16 cubes.  rowocc from launch_rowocc.
int launch_tile_order(const unsigned long long* rowocc, int total, int chunk, const TileCfg* cfg, const TileCfg* cfg_small, int n_cfg,
                      unsigned* order, unsigned* n_heavy, int tiles_cap, unsigned long long* virt, hipStream_t s);
int launch_rowocc(const float* x, unsigned long long* rowocc, int B, hipStream_t s);       // x [B][64][64][64] one channel
int launch_vrn16_row(const float* x, float* t12, float* out, const float* const* w, int B, int which, hipStream_t s, bool x_nonneg = false,
                     const RowSkip* skip = nullptr);
// Segment form of the same two kernels (vrn_seg.hip): a wave computes four SLOTS of 8 planes x 2 rows x 16 voxels taken from a
// list of the launch's heavy slots.  Every tensor the launch touches and the empty-cube responses that stand in for slots their
// producer did not write lie in one window of < 2 GiB starting at `win` (at least 2 MiB below the first tensor); *_off = byte
// offsets of cube 0 of the block input x (16 channels), of tensor1_1 | tensor2_1 (8 channels), of the block output (may equal
// x_off), and of the ONE-cube empty-cube responses of the tensor read with its halo (A: x; BC: tensor1_1 | tensor2_1) and of
// BC's residual input.  slots[i] = ((cube * 8 + plane tile) * 32 + row tile) * 4 + segment; *_virt[cube * 256 + plane tile * 32
// + row tile] bit s = that slot of the tensor was not written (nullptr: the tensor is complete).
struct SegArgs {
  const char* win = nullptr;
  unsigned x_off = 0, t_off = 0, out_off = 0, ein_off = 0, eres_off = 0;
  const unsigned* slots = nullptr;
  const unsigned* n_slots = nullptr;
  const unsigned char* in_virt = nullptr;
  const unsigned char* res_virt = nullptr;
  const float *w11 = nullptr, *b11 = nullptr, *w21 = nullptr, *b21 = nullptr, *w12 = nullptr, *b12 = nullptr, *w22 = nullptr, *b22 = nullptr,
              *w23 = nullptr, *b23 = nullptr;
};
int launch_vrn16_seg(const SegArgs& a, int which, bool x_nonneg, int max_slots, hipStream_t s);
// conv_in on slots: x = the chunk's occupancy cubes [n][64][64][64] (its own allocation), y = the stage tensor inside the window
struct ConvInSegArgs {
  const float* x = nullptr;
  const char* win = nullptr;
  unsigned out_off = 0;
  const unsigned* slots = nullptr;
  const unsigned* n_slots = nullptr;
  const float* w = nullptr;
  const float* bias = nullptr;
  int relu = 0;
};
int launch_conv_in_seg(const ConvInSegArgs& a, int max_slots, hipStream_t s);
constexpr int kSegMaxChunk = 48;                // cubes per launch of the segment form (slot codes hold the cube above bit 10)
constexpr int kSegLaunches = 7;                 // conv_in, kernel A / BC of the three C = 16 blocks
// voxel occupancy words of B cubes (occ[(b * 64 + d) * 64 + h] bit w) + the row words launch_rowocc writes; slot lists, counts
// and "not written" tables of every chunk for the kSegLaunches launches (vrn_seg.hip: seg_order_kernel)
int launch_voxocc(const float* x, unsigned long long* occ, unsigned long long* rowocc, int B, hipStream_t s);
int launch_seg_order(const unsigned long long* occ, const unsigned long long* rowocc, int total, int chunk, unsigned* slots, unsigned* counts,
                     unsigned char* virt, unsigned* counter, hipStream_t s);
// a 16-channel 64^3 tensor some of whose slots were not written, for a ROW kernel that reads it (down_1): the tensor and the
// one-cube response that stands in lie in one window (SegArgs), virt as SegArgs::in_virt
struct SegRead {
  const char* win = nullptr;
  unsigned x_off = 0, e_off = 0;
  const unsigned char* virt = nullptr;
};
// the block on NDHWC tensors for the training step: keeps tensor1_1, tensor2_1, tensor2_2 and the pre-residual output
int launch_vrn16_bwd_tail(const float* dz12, const float* dz23, const float* t11, const float* t21, const float* t22, const float* w12,
                          const float* w22, const float* w23, float* dt11, float* dt21, float* dt22, int B, hipStream_t s);
// q4: the 16-channel (and the split's 8-channel) tensors are in the Q4 layout instead of NDHWC
int launch_vrn16_bwd_tail_split(const float* dout, const int* signs, const float* t11, const float* t21, const float* t22,
                                const float* w12, const float* w22, const float* w23, float* dz12, float* dz23, float* dt11, float* dt21,
                                float* dt22, int B, hipStream_t s, bool q4 = false);
int launch_vrn16_bwd_input(const float* dt11, const float* dt21, const float* dpre, const float* x, const float* w11, const float* w21,
                           float* dx, int B, hipStream_t s, bool q4 = false);
#ifdef PCGC_EXPERIMENTS
extern int g_vrn16_abl;   // memory-ablation switches of the 64^3 row kernels, honoured in -DPCGC_EXPERIMENTS builds only
#endif
int launch_vrn16_row_train(const float* x, float* t11, float* t21, float* t22, float* pre, float* out, const float* const* w, int B,
                           hipStream_t s, int* pre_signs = nullptr, bool q4 = false);   // pre_signs != nullptr: sign bits instead of pre; q4: x / out / pre are Q4
int launch_vrn32_bwd_input(const float* dt11, const float* dt21, const float* dpre, const float* x, const float* w11, const float* w21,
                           float* dx, int B, hipStream_t s);
int launch_vrn32_bwd_tail(const float* dz12, const float* dz23, const float* t11, const float* t21, const float* t22, const float* w12,
                          const float* w22, const float* w23, float* dt11, float* dt21, float* dt22, int B, hipStream_t s);
int launch_vrn32_row_train(const float* x, float* t11, float* t21, float* t22, float* pre, float* out, const float* const* w, int B,
                           hipStream_t s, int* pre_signs = nullptr);      // pre_signs != nullptr: sign bits instead of pre
// mask (optional, Q4 like y): y = mask > 0 ? conv : 0 — the bwd-data epilogue of deconv_out's adjoint in the training step
int launch_conv_in_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s,
                       const RowSkip* skip = nullptr, const float* mask = nullptr);
int launch_deconv_out_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s);
int launch_q4_convert(const float* src, float* dst, int B, int D, int C, int to_q4, hipStream_t s);   // NDHWC <-> Q4
// C = 32 block at D = 32 (vrn_row32.hip), tensors Q4; which / w as launch_vrn16_row
int launch_vrn32_row(const float* x, float* t12, float* out, const float* const* w, int B, int which, hipStream_t s, bool x_nonneg = false,
                     const RowSkip* skip = nullptr, const float* img = nullptr);
size_t vrn32_image_floats();
int launch_vrn32_image(const float* const* w, float* dst, hipStream_t s);
// tile geometry (rows, planes) launch_vrn32_row / launch_down1_row use for B cubes: what the tile orders must be built for
void vrn32_tile_geometry(int B, int which, int* th, int* ld);
constexpr int kDown1TileRows = 2, kDown1TilePlanes = 2;
// up_2 (transposed conv 32 -> 16, 32^3 -> 64^3) as a row kernel: x Q4 at 32^3, y Q4 at 64^3; w = the filter's LDS image,
// built once by launch_row_image (row_image_floats > 0 tells which layers have one)
size_t row_image_floats(int cin, int cout, int k, int mode);
int launch_row_image(const float* w_tf, float* dst, int mode, hipStream_t s);
// the same for up to 8 images in one launch; kind: 0 = up_2's form (Conv3DTranspose layout), 1 = down_1's (Conv3D layout)
struct RowImageJobs {
  int n = 0;
  const float* w[8];
  float* dst[8];
  int kind[8];
};
int launch_row_images(const RowImageJobs& jobs, hipStream_t s);
int launch_up2_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s, bool x_nhwc = false,
                   const float* mask = nullptr);
// hyper_row.hip: the 8^3 layers of the hyperprior networks on NDHWC tensors.  conv8: 1 launched, 0 unsupported shape
int launch_up8_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s);
int launch_down8_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s);
int launch_conv8_row(const float* x, float* y, const float* w, const float* bias, int B, int Cin, int Cout, int relu, hipStream_t s);
// up_1 (transposed conv 64 -> 32, 16^3 -> 32^3) on quad vectors (vrn_row16.hip); w_image from launch_up1_image
size_t up1_image_floats();
int launch_up1_image(const float* w_tf, float* dst, hipStream_t s);
int launch_up1_row(const float* x, float* y, const float* w_image, const float* bias, int B, int relu, hipStream_t s);
// down_2 (stride-2 conv 32 -> 64, 32^3 -> 16^3) on quad vectors (vrn_row16.hip); w_image from launch_down2_image
size_t down2_image_floats();
int launch_down2_image(const float* w_tf, float* dst, hipStream_t s);
int launch_down2_row(const float* x, float* y, const float* w_image, const float* bias, int B, int relu, hipStream_t s);
// down_1 (stride-2 conv 16 -> 32, 64^3 -> 32^3) likewise: x Q4 at 64^3, y Q4 at 32^3
int launch_down1_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s, const RowSkip* skip = nullptr,
                     bool y_nhwc = false, const float* mask = nullptr, const SegRead* seg = nullptr);
// C = 64 block at D = 16 (vrn_row16.hip): which 0 = A, 1 = B (conv1_2 half), 2 = C (conv2_2 + conv2_3 half)
int launch_vrn64_row(const float* x, float* t12, float* out, const float* const* w, int B, int which, hipStream_t s, const float* img = nullptr);
size_t vrn64_image_floats();
int launch_vrn64_image(const float* const* w, float* dst, hipStream_t s);
// pack TF-layout weights for the MFMA kernel of this shape; returns floats needed (count_only) or packs
size_t mfma_packed_floats(int Cin, int Cout, int ksize, int mode);
// train_dw.hip: tiled weight gradient of the stride-1 convs; partial = [groups][taps][Cin][Cout]
int conv_dw_tile_groups(int B, int D);
int conv_dw_tile_groups_s2(int B, int D);
// *_q4: that operand is in the Q4 layout [d][h][C/4][w][4] instead of NDHWC (the training step's 64^3 stage)
int launch_conv_dw_tile_s2(const float* fine, const float* coarse, float* partial, int B, int D, int Ca, int Cb, int with_bias,
                           hipStream_t s, int fine_q4 = 0);
int launch_conv_dw_tile(const float* x, const float* dz, float* partial, int B, int D, int Cin, int Cout, int ksize,
                        int with_bias, hipStream_t s, int x_q4 = 0, int dz_q4 = 0);
bool conv_dw_pair_supported(int D, int Cin, int Cout);
int launch_conv_dw_pair(const float* x, const float* dz3, const float* dz1, float* partial3, float* partial1, int B, int D,
                        int Cin, int Cout, int with_bias, hipStream_t s, int x_q4 = 0);
int pack_weights_mfma(const float* w_tf, float* packed, int Cin, int Cout, int ksize, int mode, hipStream_t s);
// Batched weight preparation (train_plan.hip): kind 0 = pack for the MFMA kernel of (Cin, Cout, ksize, mode), kind 1 =
// flip + transpose of a stride-1 filter (its bwd-data adjoint, TF layout).  block0 = first block of the job in the
// one launch that runs a whole table (filled by the caller: running sum of (total + 255) / 256).
struct WeightJob {
  const float* src;
  float* dst;
  int kind, Cin, Cout, ksize, mode, coutp, qd, qh, total, block0;
};
size_t make_pack_job(const float* src, float* dst, int Cin, int Cout, int ksize, int mode, WeightJob* job);   // 0 = no MFMA kernel
void make_flip_job(const float* src, float* dst, int ksize, int Cin, int Cout, WeightJob* job);
int launch_weight_jobs(const WeightJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s);

// train.hip: the final reduction of one layer's weight (kind 0) or bias (kind 1) partial sums, deferred so that a
// whole backward pass finishes in one launch (train_plan.hip)
struct FinalJob {
  const float* partial;
  float* dw;
  float* db;
  int kind, taps, Cin, Cout, transposed, nchunks, cstride, block0;
};
struct FinalJobs {
  static constexpr int kMax = 56;      // 56 * 56 B + 4 < the 4 KiB kernel-argument segment
  FinalJob j[kMax];
  int n;
};

// train_dw.hip: weight gradients recorded instead of launched (a thread's capture list) and run later, equal shapes as the
// jobs of one launch — the deferred mode of the training plan (pcgc_train_plan_defer_small)
constexpr int kDwBatchMax = 8;
struct DwCall {
  int kind;                          // 0 / 1 / 2: conv_dw_mfma_kernel<16 / 32 / 64, 1, 16>; 3 / 4: conv_dw_tile_kernel<16, 16 / 32, 1>
  const float *x, *dz;
  float* partial;
  int B, D, Cin, groups, with_bias;
};
void dw_capture_begin(std::vector<DwCall>* sink);
void dw_capture_end();
int launch_dw_calls(const std::vector<DwCall>& calls, hipStream_t s);

// train.hip internals shared with train_plan.hip
int bwd_data_impl(const float* dz, const float* kernel, const float* wt_ready, const float* packed_ready, float* dx,
                  const float* relu_mask, const float* add_to, int B, int D, int Cin, int Cout, int ksize, int stride, int transposed,
                  float* wt_scratch, float* packed_scratch, hipStream_t s, int x_q4 = 0, int dz_q4 = 0);   // x_q4: dx / mask / add_to
int launch_hyper_row_conv(const ConvArgs& a, hipStream_t s);   // hyper_row.hip: 1 launched, 0 not an 8^3 hyper layer, < 0 error
size_t bwd_weight_partial_floats(int B, int D, int Cin, int Cout, int ksize, int stride, int transposed, size_t* bias_floats);
int bwd_weight_impl(const float* x, const float* dz, float* dkernel, float* dbias, int B, int D, int Cin, int Cout, int ksize,
                    int stride, int transposed, float* partial, float* bias_partial, std::vector<FinalJob>* sink, hipStream_t s,
                    int x_q4 = 0, int dz_q4 = 0);
int launch_final_jobs(const std::vector<FinalJob>& jobs, hipStream_t s);

}  // namespace pcgc
