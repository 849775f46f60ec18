// Training plan: the per-step bookkeeping of the train_hyper step kept on the library side so that
// the ~50 convolution layers of the five sub-networks cost O(1) housekeeping launches per step instead of O(layers):
//   * pcgc_train_plan_prepare: two launches pack every layer's filter for its forward MFMA kernel, flip/transpose
//     every stride-1 filter for its bwd-data pass and pack the adjoint filters (before: one pack + one flip + one pack
//     launch per layer, ~280 launches per step);
//   * pcgc_train_conv_bwd_weight only produces the per-tile partial sums, into the plan's pool;
//     pcgc_train_plan_finish_weights reduces every layer's partial sums in one or two launches (before: one or two
//     final-reduction launches per layer, ~110 per step).
// Same index arithmetic and summation order as the per-layer entry points (pcgc_conv3d_fwd / _bwd_data_fused /
// _bwd_weight), so outputs and gradients are bit-identical to them.  The plan holds raw pointers to the caller's
// parameter and gradient tensors (the trainer's flat buffers), which must stay where they are for its lifetime.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace pcgc {

struct PlanLayer {
  pcgc_train_layer d;
  int mode;                  // forward: 0 stride-1, 1 stride-2, 2 transposed
  const float* fwd_packed;   // nullptr: no MFMA kernel takes the forward shape
  const float* wt;           // flipped + transposed filter (stride-1 layers)
  const float* bwd_packed;   // adjoint filter packed for the MFMA kernel of the bwd-data shape, or nullptr
  int x_q4 = 0, y_q4 = 0;    // pcgc_train_plan_set_layout: the layer's input / output tensor (and their gradients) are Q4
  // up_2 (transposed 3^3, 32 -> 16) and down_1 (stride-2 3^3, 16 -> 32), the resamplers at the Q4 stage boundary: LDS images of
  // the filter for the inference path's row kernels (vrn_row32.hip), rebuilt by pcgc_train_plan_prepare every step — [0] the
  // layer itself, [1] its adjoint (the OTHER kernel on the same tensor: the adjoint of a stride-2 conv is the transposed conv
  // with the same tensor and vice versa)
  float* row_img[2] = {nullptr, nullptr};
};
// PCGC_TRAIN_ROW_RESAMPLE=0: up_2 / down_1 of the Q4 training step on the implicit-GEMM kernels as before round 5 (read per
// call: tests compare the two in one process)
static bool row_resample_on() {
  const char* e = getenv("PCGC_TRAIN_ROW_RESAMPLE");
  return !(e && atoi(e) == 0);
}
static bool is_up2(const pcgc_train_layer& d) { return d.transposed && d.stride == 2 && d.ksize == 3 && d.Cin == 32 && d.Cout == 16; }
static bool is_down1(const pcgc_train_layer& d) { return !d.transposed && d.stride == 2 && d.ksize == 3 && d.Cin == 16 && d.Cout == 32; }

static size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

}  // namespace pcgc

using namespace pcgc;

struct pcgc_train_plan {
  std::vector<PlanLayer> layers;
  float* blob = nullptr;                 // packed / flipped filters of every layer
  WeightJob* jobs = nullptr;             // device table: launch 1 = [0, n1) (packs of the filters, flips), launch 2 = [n1, n1 + n2)
  int n1 = 0, blocks1 = 0, n2 = 0, blocks2 = 0;      // (packs of the adjoint filters, which read launch 1's flips)
  float* scratch = nullptr;              // fallback scratch of bwd_data_impl (unused: everything is prepared)
  size_t scratch_wt = 0;
  // weight-gradient partial sums of the running backward pass
  std::vector<float*> pools;             // pools[0] serves a step; overflow pools appear while it is too small
  size_t pool_floats = 0, pool_used = 0, pool_wanted = 0;
  std::vector<FinalJob> finals;
  // pcgc_train_plan_defer_small: weight gradients of the stride-1 layers at D <= 16 are recorded here and run by
  // pcgc_train_plan_finish_weights, equal shapes as the jobs of one launch (train_dw.hip launch_dw_calls)
  bool defer_small = false;
  std::vector<DwCall> deferred;
};

static float* pool_take(pcgc_train_plan* p, size_t floats) {
  floats = align64(floats);
  p->pool_wanted += floats;
  if (p->pool_used + floats > p->pool_floats) {       // first step at a new size: an overflow pool now, one right-sized pool next step
    float* extra = nullptr;
    if (hipMalloc((void**)&extra, floats * sizeof(float)) != hipSuccess) return nullptr;
    p->pools.push_back(extra);
    return extra;
  }
  float* r = p->pools[0] + p->pool_used;
  p->pool_used += floats;
  return r;
}

extern "C" {

void pcgc_train_plan_destroy(pcgc_train_plan* p);
static int plan_fill(pcgc_train_plan* p, const pcgc_train_layer* layers, int n_layers);

int pcgc_train_plan_create(const pcgc_train_layer* layers, int n_layers, pcgc_train_plan** out) {
  PCGC_REQUIRE(layers && out && n_layers > 0, "pcgc_train_plan_create: NULL argument");
  *out = nullptr;
  pcgc_train_plan* p = new pcgc_train_plan();
  const int rc = plan_fill(p, layers, n_layers);
  if (rc) { pcgc_train_plan_destroy(p); return rc; }        // frees whatever was allocated before the failure
  *out = p;
  return 0;
}

static int plan_fill(pcgc_train_plan* p, const pcgc_train_layer* layers, int n_layers) {
  for (int i = 0; i < n_layers; ++i) {
    const pcgc_train_layer& d = layers[i];
    PCGC_REQUIRE(d.kernel && d.dkernel && d.Cin > 0 && d.Cout > 0 && d.ksize >= 1 && d.ksize <= 9 && (d.ksize & 1) && (d.stride == 1 || d.stride == 2) &&
                     (!d.transposed || d.stride == 2) && d.Cin <= 64 && d.Cout <= 64,
                 "pcgc_train_plan_create: layer %d unsupported (Cin=%d Cout=%d k=%d stride=%d transposed=%d)", i, d.Cin, d.Cout, d.ksize,
                 d.stride, d.transposed);
  }
  p->layers.resize(n_layers);
  struct Pending { WeightJob job; int layer, what; size_t dst_off, src_off; };   // what: 0 fwd_packed, 1 wt, 2 bwd_packed
  std::vector<Pending> first, second;
  struct RowImages { int layer; size_t off0, off1; };
  std::vector<RowImages> row_images;
  size_t total = 0;
  for (int i = 0; i < n_layers; ++i) {
    const pcgc_train_layer& d = layers[i];
    PlanLayer& L = p->layers[i];
    L.d = d;
    L.mode = d.transposed ? 2 : (d.stride == 2 ? 1 : 0);
    L.fwd_packed = L.wt = L.bwd_packed = nullptr;
    const size_t wn = (size_t)d.ksize * d.ksize * d.ksize * d.Cin * d.Cout;
    p->scratch_wt = std::max(p->scratch_wt, align64(wn));
    WeightJob j;
    size_t n = make_pack_job(d.kernel, nullptr, d.Cin, d.Cout, d.ksize, L.mode, &j);
    if (n) { first.push_back({j, i, 0, total, 0}); total += align64(n); }
    size_t wt_off = (size_t)-1;
    if (L.mode == 0) {                   // adjoint of a stride-1 conv: the flipped filter as a stride-1 conv Cout -> Cin
      make_flip_job(d.kernel, nullptr, d.ksize, d.Cin, d.Cout, &j);
      wt_off = total;
      first.push_back({j, i, 1, total, 0});
      total += align64(wn);
    }
    // adjoint of the stride-2 conv = transposed conv with the same tensor, and vice versa (train.hip bwd_data_impl)
    const int bmode = L.mode == 0 ? 0 : (L.mode == 1 ? 2 : 1);
    n = make_pack_job(nullptr, nullptr, d.Cout, d.Cin, d.ksize, bmode, &j);
    if (n) { second.push_back({j, i, 2, total, wt_off}); total += align64(n); }
    if (is_up2(d) || is_down1(d)) {      // the two LDS images for the row kernels (offsets now, pointers once the blob exists)
      const size_t img = align64(row_image_floats(d.Cin, d.Cout, d.ksize, L.mode));
      row_images.push_back({i, total, total + img});
      total += 2 * img;
    }
  }
  size_t scratch_packed = 0;
  for (int i = 0; i < n_layers; ++i)
    for (int m = 0; m < 3; ++m) scratch_packed = std::max(scratch_packed, mfma_packed_floats(layers[i].Cout, layers[i].Cin, layers[i].ksize, m));
  PCGC_CHECK_HIP(hipMalloc((void**)&p->blob, std::max<size_t>(total, 64) * sizeof(float)));
  PCGC_CHECK_HIP(hipMalloc((void**)&p->scratch, (p->scratch_wt + scratch_packed + 64) * sizeof(float)));
  for (const RowImages& r : row_images) {
    p->layers[r.layer].row_img[0] = p->blob + r.off0;
    p->layers[r.layer].row_img[1] = p->blob + r.off1;
  }
  std::vector<WeightJob> table;
  for (int pass = 0; pass < 2; ++pass) {
    int blocks = 0;
    for (Pending& q : pass == 0 ? first : second) {
      q.job.dst = p->blob + q.dst_off;
      if (pass == 1) q.job.src = q.src_off == (size_t)-1 ? layers[q.layer].kernel : p->blob + q.src_off;
      q.job.block0 = blocks;
      blocks += (q.job.total + 255) / 256;
      PlanLayer& L = p->layers[q.layer];
      (q.what == 0 ? L.fwd_packed : (q.what == 1 ? L.wt : L.bwd_packed)) = q.job.dst;
      table.push_back(q.job);
    }
    (pass == 0 ? p->blocks1 : p->blocks2) = blocks;
  }
  p->n1 = (int)first.size();
  p->n2 = (int)second.size();
  PCGC_CHECK_HIP(hipMalloc((void**)&p->jobs, std::max<size_t>(table.size(), 1) * sizeof(WeightJob)));
  if (!table.empty()) PCGC_CHECK_HIP(hipMemcpy(p->jobs, table.data(), table.size() * sizeof(WeightJob), hipMemcpyHostToDevice));
  p->pools.push_back(nullptr);
  p->finals.reserve(2 * (size_t)n_layers);
  return 0;
}

void pcgc_train_plan_destroy(pcgc_train_plan* p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  for (float* q : p->pools)
    if (q) (void)hipFree(q);
  if (p->blob) (void)hipFree(p->blob);
  if (p->scratch) (void)hipFree(p->scratch);
  if (p->jobs) (void)hipFree(p->jobs);
  delete p;
}

int pcgc_train_plan_layers(const pcgc_train_plan* p) { return p ? (int)p->layers.size() : 0; }

/* Start of a step: refresh every prepared filter from the current parameter values and reset the partial-sum pool. */
int pcgc_train_plan_prepare(pcgc_train_plan* p, pcgc_stream_t stream) {
  PCGC_REQUIRE(p, "pcgc_train_plan_prepare: NULL plan");
  hipStream_t s = (hipStream_t)stream;
  if (p->pools.size() > 1 || p->pool_wanted > p->pool_floats) {       // the last step overflowed: one pool of the right size
    PCGC_CHECK_HIP(hipStreamSynchronize(s));
    for (float* q : p->pools)
      if (q) PCGC_CHECK_HIP(hipFree(q));
    p->pools.assign(1, nullptr);
    p->pool_floats = p->pool_wanted;
    PCGC_CHECK_HIP(hipMalloc((void**)&p->pools[0], std::max<size_t>(p->pool_floats, 64) * sizeof(float)));
  }
  p->pool_used = p->pool_wanted = 0;
  p->finals.clear();
  p->deferred.clear();
  int rc = launch_weight_jobs(p->jobs, p->n1, p->blocks1, s);
  if (rc) return rc;
  if ((rc = launch_weight_jobs(p->jobs + p->n1, p->n2, p->blocks2, s))) return rc;
  RowImageJobs imgs;                     // the resamplers' LDS images (the layer's own form and its adjoint's), one launch
  for (const PlanLayer& L : p->layers)
    if (L.row_img[0]) {
      if (imgs.n + 2 > 8) { if ((rc = launch_row_images(imgs, s))) return rc; imgs.n = 0; }
      imgs.w[imgs.n] = L.d.kernel; imgs.dst[imgs.n] = L.row_img[0]; imgs.kind[imgs.n] = L.mode == 2 ? 0 : 1; ++imgs.n;
      imgs.w[imgs.n] = L.d.kernel; imgs.dst[imgs.n] = L.row_img[1]; imgs.kind[imgs.n] = L.mode == 2 ? 1 : 0; ++imgs.n;
    }
  return launch_row_images(imgs, s);
}

/* Layout of one layer's tensors for every later call on it: x_q4 / y_q4 != 0 = the layer's input / output (and the
 * gradients laid out like them) are Q4 [b][d][h][C/4][w][4] instead of NDHWC.  The training step keeps the 16- and
 * 8-channel tensors of its 64^3 stage that way (the row kernels' native layout: 1 KiB per wave instruction instead of
 * 16 B per lane at a 64 B stride); only the shapes of that stage have kernels that read it — others fail loudly. */
int pcgc_train_plan_set_layout(pcgc_train_plan* p, int layer, int x_q4, int y_q4) {
  PCGC_REQUIRE(p && layer >= 0 && layer < (int)p->layers.size(), "pcgc_train_plan_set_layout: bad argument");
  p->layers[layer].x_q4 = x_q4 != 0;
  p->layers[layer].y_q4 = y_q4 != 0;
  return 0;
}

/* Forward of layer `layer` (pcgc_conv3d_fwd with algo 0 on the prepared filter). */
int pcgc_train_conv_fwd(const pcgc_train_plan* p, int layer, const float* x, const float* bias, float* y, int B, int D, int relu,
                        pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer >= 0 && layer < (int)p->layers.size() && x && y, "pcgc_train_conv_fwd: bad argument");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const PlanLayer& L = p->layers[layer];
  PCGC_REQUIRE(L.mode != 1 || D % 2 == 0, "pcgc_train_conv_fwd: stride-2 conv needs even D");
  ConvArgs a;
  a.x = x; a.w = L.d.kernel; a.bias = bias; a.y = y; a.res = nullptr;
  a.B = B; a.Din = D; a.Dout = L.mode == 2 ? 2 * D : (L.mode == 1 ? D / 2 : D);
  a.Cin = L.d.Cin; a.Cout = L.d.Cout; a.x_cs = L.d.Cin; a.x_co = 0; a.y_cs = L.d.Cout; a.y_co = 0;
  a.ksize = L.d.ksize; a.mode = L.mode; a.relu = relu; a.absval = 0; a.lower_bound = 0.f;
  a.w2 = nullptr; a.bias2 = nullptr; a.y2 = nullptr; a.y2_cs = 0; a.cout2 = 0;
  if (L.x_q4 || L.y_q4) {
    // Q4 stage boundary: conv_in / deconv_out on the inference path's row kernels, down_1 / up_2 on the MFMA kernels
    // that take the flags; nothing else may be handed a Q4 tensor
    a.x_q4 = L.x_q4; a.y_q4 = L.y_q4;
    if (L.d.Cin == 1 && L.d.Cout == 16 && L.d.ksize == 3 && L.mode == 0 && D == 64 && L.y_q4 && !L.x_q4)
      return launch_conv_in_row(x, y, L.d.kernel, bias, B, relu, s);
    if (L.d.Cin == 16 && L.d.Cout == 1 && L.d.ksize == 3 && L.mode == 0 && D == 64 && L.x_q4 && !L.y_q4)
      return launch_deconv_out_row(x, y, L.d.kernel, bias, B, relu, s);
    // up_2 / down_1 between the NDHWC 32^3 stage and the Q4 64^3 stage: the inference path's row kernels (84 / 78 us per 8 cubes
    // against 140 / 120 us for the implicit-GEMM kernels)
    if (L.row_img[0] && row_resample_on() && is_up2(L.d) && D == 32 && !L.x_q4 && L.y_q4) return launch_up2_row(x, y, L.row_img[0], bias, B, relu, s, true);
    if (L.row_img[0] && row_resample_on() && is_down1(L.d) && D == 64 && L.x_q4 && !L.y_q4) return launch_down1_row(x, y, L.row_img[0], bias, B, relu, s, nullptr, true);
    if (L.mode != 0 && L.fwd_packed && launch_conv_mfma(a, nullptr, s, false) == 1) {
      const int rc = launch_conv_mfma(a, L.fwd_packed, s, true);
      return rc < 0 ? rc : 0;
    }
    set_error("pcgc_train_conv_fwd: layer %d has no kernel for Q4 tensors (Cin=%d Cout=%d k=%d mode=%d D=%d)", layer, L.d.Cin, L.d.Cout,
              L.d.ksize, L.mode, D);
    return -1;
  }
  if (L.d.Cin == 1 || L.d.Cout == 1) {                      // conv_in / deconv_out: the LDS-tiled VALU kernel (as pcgc_net_forward)
    const int rc = launch_conv_valu(a, s, true);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (L.fwd_packed && launch_conv_mfma(a, nullptr, s, false) == 1) {
    const int rc = launch_conv_mfma(a, L.fwd_packed, s, true);
    return rc < 0 ? rc : 0;
  }
  if (const int rc = launch_hyper_row_conv(a, s)) return rc < 0 ? rc : 0;      // the 8^3 hyper layers: row kernels
  return launch_conv_direct(a, s);
}

/* Gradient w.r.t. the layer's input (pcgc_conv3d_bwd_data_fused on the prepared adjoint filter). */
int pcgc_train_conv_bwd_data(const pcgc_train_plan* p, int layer, const float* dz, float* dx, const float* relu_mask,
                             const float* add_to, int B, int D, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer >= 0 && layer < (int)p->layers.size() && dz && dx, "pcgc_train_conv_bwd_data: bad argument");
  const PlanLayer& L = p->layers[layer];
  if (B == 0) return 0;
  // the reverse of down_1 is up_2's kernel on down_1's tensor (dz NDHWC 32^3 -> dx Q4 64^3), the reverse of up_2 is down_1's
  // kernel on up_2's tensor (dz Q4 64^3 -> dx NDHWC 32^3); the ReLU mask of the layer's input rides in the store
  if (L.row_img[1] && row_resample_on() && !add_to && is_down1(L.d) && D == 64 && L.x_q4 && !L.y_q4)
    return launch_up2_row(dz, dx, L.row_img[1], nullptr, B, 0, (hipStream_t)stream, true, relu_mask);
  if (L.row_img[1] && row_resample_on() && !add_to && is_up2(L.d) && D == 32 && !L.x_q4 && L.y_q4)
    return launch_down1_row(dz, dx, L.row_img[1], nullptr, B, 0, (hipStream_t)stream, nullptr, true, relu_mask);
  return bwd_data_impl(dz, L.d.kernel, L.wt, L.bwd_packed, dx, relu_mask, add_to, B, D, L.d.Cin, L.d.Cout, L.d.ksize, L.d.stride,
                       L.d.transposed, p->scratch, p->scratch + p->scratch_wt, (hipStream_t)stream, L.x_q4, L.y_q4);
}

/* Two stride-1 layers of a 16^3 VRN block in one launch (conv_mfma_pair_kernel): conv1_1 | conv2_1 (same input),
 * conv1_2 | conv2_2 (independent tensors).  Shapes without a pair kernel run as the two single calls; same bits either way. */
int pcgc_train_conv_fwd_pair(const pcgc_train_plan* p, int layer_a, int layer_b, const float* xa, const float* xb, const float* bias_a,
                             const float* bias_b, float* ya, float* yb, int B, int D, int relu_a, int relu_b, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer_a >= 0 && layer_a < (int)p->layers.size() && layer_b >= 0 && layer_b < (int)p->layers.size() && xa && xb && ya && yb,
               "pcgc_train_conv_fwd_pair: bad argument");
  if (B == 0) return 0;
  const PlanLayer& La = p->layers[layer_a];
  const PlanLayer& Lb = p->layers[layer_b];
  if (La.mode == 0 && Lb.mode == 0 && La.fwd_packed && Lb.fwd_packed && !La.x_q4 && !La.y_q4 && !Lb.x_q4 && !Lb.y_q4) {
    ConvArgs a[2];
    const PlanLayer* L[2] = {&La, &Lb};
    const float* x[2] = {xa, xb};
    const float* bias[2] = {bias_a, bias_b};
    float* y[2] = {ya, yb};
    const int relu[2] = {relu_a, relu_b};
    for (int i = 0; i < 2; ++i) {
      ConvArgs& c = a[i];
      c.x = x[i]; c.w = nullptr; c.bias = bias[i]; c.y = y[i]; c.res = nullptr;
      c.B = B; c.Din = D; c.Dout = D;
      c.Cin = L[i]->d.Cin; c.Cout = L[i]->d.Cout; c.x_cs = c.Cin; c.x_co = 0; c.y_cs = c.Cout; c.y_co = 0;
      c.ksize = L[i]->d.ksize; c.mode = 0; c.relu = relu[i]; c.absval = 0; c.lower_bound = 0.f;
      c.w2 = nullptr; c.bias2 = nullptr; c.y2 = nullptr; c.y2_cs = 0; c.cout2 = 0;
    }
    const int rc = launch_conv_mfma_pair(a[0], La.fwd_packed, a[1], Lb.fwd_packed, (hipStream_t)stream);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (const int rc = pcgc_train_conv_fwd(p, layer_a, xa, bias_a, ya, B, D, relu_a, stream)) return rc;
  return pcgc_train_conv_fwd(p, layer_b, xb, bias_b, yb, B, D, relu_b, stream);
}

/* The reverse of two stride-1 layers in one launch (conv1_2^T | conv2_3^T of a 16^3 block): dx_i = (mask_i > 0) * conv_i^T(dz_i),
 * nothing accumulated.  Shapes without a pair kernel run as two pcgc_train_conv_bwd_data calls; same bits either way. */
int pcgc_train_conv_bwd_data_pair(const pcgc_train_plan* p, int layer_a, int layer_b, const float* dz_a, const float* dz_b, float* dx_a,
                                  float* dx_b, const float* relu_mask_a, const float* relu_mask_b, int B, int D, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer_a >= 0 && layer_a < (int)p->layers.size() && layer_b >= 0 && layer_b < (int)p->layers.size() && dz_a && dz_b && dx_a &&
                   dx_b, "pcgc_train_conv_bwd_data_pair: bad argument");
  if (B == 0) return 0;
  const PlanLayer& La = p->layers[layer_a];
  const PlanLayer& Lb = p->layers[layer_b];
  if (La.mode == 0 && Lb.mode == 0 && La.bwd_packed && Lb.bwd_packed && !La.x_q4 && !La.y_q4 && !Lb.x_q4 && !Lb.y_q4) {
    ConvArgs a[2];
    const PlanLayer* L[2] = {&La, &Lb};
    const float* dz[2] = {dz_a, dz_b};
    float* dx[2] = {dx_a, dx_b};
    const float* mask[2] = {relu_mask_a, relu_mask_b};
    for (int i = 0; i < 2; ++i) {                       // as bwd_data_impl builds them: the forward kernel on the adjoint filter
      ConvArgs& c = a[i];
      c.x = dz[i]; c.w = nullptr; c.bias = nullptr; c.y = dx[i]; c.res = nullptr;
      c.B = B; c.Din = D; c.Dout = D;
      c.Cin = L[i]->d.Cout; c.Cout = L[i]->d.Cin; c.x_cs = c.Cin; c.x_co = 0; c.y_cs = c.Cout; c.y_co = 0;
      c.ksize = L[i]->d.ksize; c.mode = 0; c.relu = 0; c.absval = 0; c.lower_bound = 0.f;
      c.w2 = nullptr; c.bias2 = nullptr; c.y2 = nullptr; c.y2_cs = 0; c.cout2 = 0;
      c.mask = mask[i]; c.add_to = nullptr;
    }
    const int rc = launch_conv_mfma_pair(a[0], La.bwd_packed, a[1], Lb.bwd_packed, (hipStream_t)stream);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (const int rc = pcgc_train_conv_bwd_data(p, layer_a, dz_a, dx_a, relu_mask_a, nullptr, B, D, stream)) return rc;
  return pcgc_train_conv_bwd_data(p, layer_b, dz_b, dx_b, relu_mask_b, nullptr, B, D, stream);
}

/* The reverse of a block's two input layers in one launch (conv_mfma_chain_kernel): dx = m * (m * (dx + conv_a^T(dz_a)) + conv_b^T(dz_b))
 * in place, m = (relu_mask > 0) or 1; layer b 1x1x1.  Other shapes: the two pcgc_train_conv_bwd_data calls with add_to = dx. */
int pcgc_train_conv_bwd_data_chain(const pcgc_train_plan* p, int layer_a, int layer_b, const float* dz_a, const float* dz_b, float* dx,
                                   const float* relu_mask, int B, int D, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer_a >= 0 && layer_a < (int)p->layers.size() && layer_b >= 0 && layer_b < (int)p->layers.size() && dz_a && dz_b && dx,
               "pcgc_train_conv_bwd_data_chain: bad argument");
  if (B == 0) return 0;
  const PlanLayer& La = p->layers[layer_a];
  const PlanLayer& Lb = p->layers[layer_b];
  if (La.mode == 0 && Lb.mode == 0 && La.bwd_packed && Lb.bwd_packed && !La.x_q4 && !La.y_q4 && !Lb.x_q4 && !Lb.y_q4 && Lb.d.ksize == 1 &&
      La.d.Cin == Lb.d.Cin) {
    ConvArgs a[2];
    const PlanLayer* L[2] = {&La, &Lb};
    const float* dz[2] = {dz_a, dz_b};
    for (int i = 0; i < 2; ++i) {                       // as bwd_data_impl builds them
      ConvArgs& c = a[i];
      c.x = dz[i]; c.w = nullptr; c.bias = nullptr; c.y = dx; c.res = nullptr;
      c.B = B; c.Din = D; c.Dout = D;
      c.Cin = L[i]->d.Cout; c.Cout = L[i]->d.Cin; c.x_cs = c.Cin; c.x_co = 0; c.y_cs = c.Cout; c.y_co = 0;
      c.ksize = L[i]->d.ksize; c.mode = 0; c.relu = 0; c.absval = 0; c.lower_bound = 0.f;
      c.w2 = nullptr; c.bias2 = nullptr; c.y2 = nullptr; c.y2_cs = 0; c.cout2 = 0;
      c.mask = relu_mask; c.add_to = dx;
    }
    const int rc = launch_conv_mfma_chain(a[0], La.bwd_packed, a[1], Lb.bwd_packed, (hipStream_t)stream);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (const int rc = pcgc_train_conv_bwd_data(p, layer_a, dz_a, dx, relu_mask, dx, B, D, stream)) return rc;
  return pcgc_train_conv_bwd_data(p, layer_b, dz_b, dx, relu_mask, dx, B, D, stream);
}

/* The last layer of a block's second path (1x1x1: y = t23) and the block's merge out = relu(blk_x + [t12 | t23]) in one launch
 * (conv_mfma_merge_kernel); other shapes: pcgc_train_conv_fwd, then pcgc_vrn_merge.  C = channels of the block. */
int pcgc_train_conv_fwd_merge(const pcgc_train_plan* p, int layer, const float* x, const float* bias, float* y, int relu, const float* blk_x,
                              const float* t12, float* out, int C, int B, int D, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer >= 0 && layer < (int)p->layers.size() && x && y && blk_x && t12 && out && C > 0 && C % 8 == 0,
               "pcgc_train_conv_fwd_merge: bad argument");
  if (B == 0) return 0;
  const PlanLayer& L = p->layers[layer];
  PCGC_REQUIRE(L.mode == 0 && 2 * L.d.Cout == C, "pcgc_train_conv_fwd_merge: the layer must write half the block's channels at the block's size");
  if (L.fwd_packed && !L.x_q4 && !L.y_q4) {
    ConvArgs c;
    c.x = x; c.w = nullptr; c.bias = bias; c.y = y; c.res = nullptr;
    c.B = B; c.Din = D; c.Dout = D;
    c.Cin = L.d.Cin; c.Cout = L.d.Cout; c.x_cs = c.Cin; c.x_co = 0; c.y_cs = c.Cout; c.y_co = 0;
    c.ksize = L.d.ksize; c.mode = 0; c.relu = relu; c.absval = 0; c.lower_bound = 0.f;
    c.w2 = nullptr; c.bias2 = nullptr; c.y2 = nullptr; c.y2_cs = 0; c.cout2 = 0;
    const MergeArgs m{blk_x, t12, out, C};
    const int rc = launch_conv_mfma_merge(c, L.fwd_packed, m, (hipStream_t)stream);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (const int rc = pcgc_train_conv_fwd(p, layer, x, bias, y, B, D, relu, stream)) return rc;
  return pcgc_vrn_merge(blk_x, t12, y, out, (int64_t)B * D * D * D, C, stream);
}

/* Partial sums of the layer's weight (and bias) gradient; dkernel / dbias are written by pcgc_train_plan_finish_weights. */
int pcgc_train_conv_bwd_weight(pcgc_train_plan* p, int layer, const float* x, const float* dz, int B, int D, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer >= 0 && layer < (int)p->layers.size() && x && dz, "pcgc_train_conv_bwd_weight: bad argument");
  if (B == 0) return 0;
  const PlanLayer& L = p->layers[layer];
  size_t bias_floats = 0;
  const size_t n = bwd_weight_partial_floats(B, D, L.d.Cin, L.d.Cout, L.d.ksize, L.d.stride, L.d.transposed, &bias_floats);
  float* partial = pool_take(p, n);
  float* bp = L.d.dbias ? pool_take(p, bias_floats) : nullptr;
  PCGC_REQUIRE(partial && (bp || !L.d.dbias), "pcgc_train_conv_bwd_weight: out of device memory for the partial sums");
  const bool defer = p->defer_small && D <= 16 && L.mode == 0 && !L.x_q4 && !L.y_q4;
  if (defer) dw_capture_begin(&p->deferred);       // the batchable shapes are recorded, anything else launches as always
  const int rc = bwd_weight_impl(x, dz, L.d.dkernel, L.d.dbias, B, D, L.d.Cin, L.d.Cout, L.d.ksize, L.d.stride, L.d.transposed, partial, bp,
                                 &p->finals, (hipStream_t)stream, L.x_q4, L.y_q4);
  if (defer) dw_capture_end();
  return rc;
}

/* on != 0: weight gradients of the stride-1 layers at D <= 16 (the 16^3 stage and the hyperprior nets of the training step:
 * ~35 launches of 128-512 workgroups per step) are not launched by pcgc_train_conv_bwd_weight but by
 * pcgc_train_plan_finish_weights, equal shapes as the jobs of one launch.  The caller then keeps x and dz of those calls
 * alive and unchanged until pcgc_train_plan_finish_weights.  Same kernels, same sums: bit-identical gradients. */
int pcgc_train_plan_defer_small(pcgc_train_plan* p, int on) {
  PCGC_REQUIRE(p, "pcgc_train_plan_defer_small: NULL plan");
  PCGC_REQUIRE(p->deferred.empty(), "pcgc_train_plan_defer_small: weight gradients are pending (call it between steps)");
  p->defer_small = on != 0;
  return 0;
}

/* The two layers of a VRN block that read the block input — `layer3` (3x3x3) and `layer1` (1x1x1), same Cin -> Cout — in one
 * pass over x where the fused kernel exists (16 | Cin, Cout 4 or 8, biases on both); otherwise the two single calls. */
int pcgc_train_conv_bwd_weight_pair(pcgc_train_plan* p, int layer3, int layer1, const float* x, const float* dz3, const float* dz1,
                                    int B, int D, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && layer3 >= 0 && layer3 < (int)p->layers.size() && layer1 >= 0 && layer1 < (int)p->layers.size() && x && dz3 && dz1,
               "pcgc_train_conv_bwd_weight_pair: bad argument");
  if (B == 0) return 0;
  const PlanLayer& L3 = p->layers[layer3];
  const PlanLayer& L1 = p->layers[layer1];
  const bool shape_ok = L3.d.ksize == 3 && L1.d.ksize == 1 && L3.d.stride == 1 && L1.d.stride == 1 && !L3.d.transposed &&
                        !L1.d.transposed && L3.d.Cin == L1.d.Cin && L3.d.Cout == L1.d.Cout && L3.d.dbias && L1.d.dbias &&
                        256 % L3.d.Cout == 0 && conv_dw_pair_supported(D, L3.d.Cin, L3.d.Cout);
  if (shape_ok) {
    size_t bf3 = 0, bf1 = 0;
    const size_t n3 = bwd_weight_partial_floats(B, D, L3.d.Cin, L3.d.Cout, 3, 1, 0, &bf3);
    const size_t n1 = bwd_weight_partial_floats(B, D, L1.d.Cin, L1.d.Cout, 1, 1, 0, &bf1);
    float* p3 = pool_take(p, n3);
    float* p1 = pool_take(p, n1);
    PCGC_REQUIRE(p3 && p1, "pcgc_train_conv_bwd_weight_pair: out of device memory for the partial sums");
    const int rc = launch_conv_dw_pair(x, dz3, dz1, p3, p1, B, D, L3.d.Cin, L3.d.Cout, 1, (hipStream_t)stream, L3.x_q4);
    if (rc != 1) return rc < 0 ? rc : -1;
    const int groups = conv_dw_tile_groups(B, D), co = L3.d.Cout, ci = L3.d.Cin;
    p->finals.push_back(FinalJob{p3, L3.d.dkernel, L3.d.dbias, 0, 27, ci, co, 0, groups, 27 * ci * co + co, 0});
    p->finals.push_back(FinalJob{p1, L1.d.dkernel, L1.d.dbias, 0, 1, ci, co, 0, groups, ci * co + co, 0});
    return 0;
  }
  int rc = pcgc_train_conv_bwd_weight(p, layer3, x, dz3, B, D, stream);
  if (rc) return rc;
  return pcgc_train_conv_bwd_weight(p, layer1, x, dz1, B, D, stream);
}

/* End of the backward pass: every pending final reduction, in one launch per 56 jobs. */
int pcgc_train_plan_finish_weights(pcgc_train_plan* p, pcgc_stream_t stream) {
  PCGC_REQUIRE(p, "pcgc_train_plan_finish_weights: NULL plan");
  int rc = p->deferred.empty() ? 0 : launch_dw_calls(p->deferred, (hipStream_t)stream);
  p->deferred.clear();
  if (!rc) rc = launch_final_jobs(p->finals, (hipStream_t)stream);
  p->finals.clear();
  return rc;
}

}  // extern "C"
