// Whole-transform executors behind pcgc_net_* (include/pcgc.h).
//
// Layer tables restate models/model_voxception.py (AnalysisTransform 71-144,
// SynthesisTransform 147-214, HyperEncoder 217-252, HyperDecoder 255-308,
// _VoxceptionResNet 11-68); they must match pcgcv1_amd/models/spec.py, which the
// Python host uses to order the parameter list.
//
// The reference runs one cube per call (tf.map_fn, transform.py:116-147, 224-257);
// here a batch is cut into chunks of cubes small enough that the chunk's working
// set stays in the 256 MiB Infinity Cache and large enough to fill 256 CUs, and
// every layer is one launch over the whole chunk.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace pcgc {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct LayerDef {
  const char* name;
  int tconv, cin, cout, k, stride, bias, relu;
};

static void push_vrn(std::vector<LayerDef>& L, int c) {
  const int q = c / 4, h = c / 2;
  L.push_back({"conv1_1", 0, c, q, 3, 1, 1, 1});
  L.push_back({"conv1_2", 0, q, h, 3, 1, 1, 1});
  L.push_back({"conv2_1", 0, c, q, 1, 1, 1, 1});
  L.push_back({"conv2_2", 0, q, q, 3, 1, 1, 1});
  L.push_back({"conv2_3", 0, q, h, 1, 1, 1, 1});
}

static std::vector<LayerDef> layer_table(int kind) {
  std::vector<LayerDef> L;
  switch (kind) {
    case PCGC_NET_ANALYSIS:
      L.push_back({"conv_in", 0, 1, 16, 3, 1, 1, 1});
      for (int i = 0; i < 3; ++i) push_vrn(L, 16);
      L.push_back({"down_1", 0, 16, 32, 3, 2, 0, 1});
      for (int i = 0; i < 3; ++i) push_vrn(L, 32);
      L.push_back({"down_2", 0, 32, 64, 3, 2, 0, 1});
      for (int i = 0; i < 3; ++i) push_vrn(L, 64);
      L.push_back({"conv_out", 0, 64, 16, 3, 1, 1, 0});
      break;
    case PCGC_NET_SYNTHESIS:
      L.push_back({"deconv_in", 0, 16, 64, 3, 1, 1, 1});
      for (int i = 0; i < 3; ++i) push_vrn(L, 64);
      L.push_back({"up_1", 1, 64, 32, 3, 2, 1, 1});
      for (int i = 0; i < 3; ++i) push_vrn(L, 32);
      L.push_back({"up_2", 1, 32, 16, 3, 2, 1, 1});
      for (int i = 0; i < 3; ++i) push_vrn(L, 16);
      L.push_back({"deconv_out", 0, 16, 1, 3, 1, 1, 0});
      break;
    case PCGC_NET_HYPER_ENCODER:
      L.push_back({"conv1", 0, 16, 16, 3, 1, 1, 1});
      L.push_back({"conv2", 0, 16, 16, 3, 2, 1, 1});
      L.push_back({"conv3", 0, 16, 8, 3, 1, 1, 0});
      break;
    case PCGC_NET_HYPER_DECODER:
      L.push_back({"conv1", 0, 8, 16, 3, 1, 1, 1});
      L.push_back({"conv2", 1, 16, 16, 3, 2, 1, 1});
      L.push_back({"conv3", 0, 16, 32, 3, 1, 1, 1});
      L.push_back({"conv4_1", 0, 32, 16, 3, 1, 1, 0});
      L.push_back({"conv4_2", 0, 32, 16, 3, 1, 1, 0});
      break;
    default:
      break;
  }
  return L;
}

struct LayerW {
  LayerDef def;
  const float* w_tf;    // device, TF layout
  const float* bias;    // device or nullptr
  const float* w_mfma;  // device, packed (nullptr when no MFMA kernel takes this shape at any size)
  const float* w_row = nullptr;   // device, LDS image of the row kernel that takes this layer (up_2, down_1), or nullptr
};

}  // namespace pcgc

struct ProfRec {
  int layer, mfma, B, Din;
  hipEvent_t t0, t1;
};

struct pcgc_net {
  int kind;
  int algo;          // 0 auto, 1 direct only
  int chunk;         // cubes per chunk
  std::vector<pcgc::LayerW> layers;
  float* blob;       // all weights (TF + packed), library-owned device memory
  // analysis only: the 64^3 stage's response to an EMPTY cube (RowSkip, common.h) — conv_in's output, then tensor1_1 |
  // tensor2_1 and the output of each of the three C = 16 blocks, Q4, one cube each; library-owned like the weights
  float* empty_blob = nullptr;
  const float* E_in = nullptr;
  const float* E_t[3] = {nullptr, nullptr, nullptr};
  const float* E_o[3] = {nullptr, nullptr, nullptr};
  const pcgc::TileCfg* skip_cfg = nullptr;  // device table of the kSkipLaunches launch geometries of the 64^3 stage (in empty_blob)
  const pcgc::TileCfg* skip_cfg_mid[2] = {nullptr, nullptr};   // ... of the 32^3 stage's six launches: [0] large launches, [1] <= 16 cubes
  const float* E_d1 = nullptr;              // down_1's and the three C = 32 blocks' responses to an empty cube (32^3)
  const float* E_t32[3] = {nullptr, nullptr, nullptr};
  const float* E_o32[3] = {nullptr, nullptr, nullptr};
  unsigned* skip_counter = nullptr;   // tests: device word that counts the wave tiles skipped (pcgc_net_set_skip_counter)
  bool profiling = false;
  mutable std::vector<ProfRec> prof;
};

namespace pcgc {

static inline int mode_of(const LayerDef& d) { return d.tconv ? 2 : (d.stride == 2 ? 1 : 0); }

struct Exec {
  const pcgc_net* net;
  hipStream_t s;
  int B;  // cubes in this chunk

  ConvArgs args(const LayerW& L, const float* x, int Din, int x_cs, int x_co, float* y, int y_cs, int y_co,
                const float* res, int absval = 0, float lb = 0.f) const {
    ConvArgs a;
    a.x = x; a.w = L.w_tf; a.bias = L.bias; a.y = y; a.res = res;
    a.B = B; a.Din = Din;
    a.Dout = L.def.tconv ? Din * 2 : Din / L.def.stride;
    a.Cin = L.def.cin; a.Cout = L.def.cout;
    a.x_cs = x_cs; a.x_co = x_co; a.y_cs = y_cs; a.y_co = y_co;
    a.ksize = L.def.k; a.mode = mode_of(L.def); a.relu = L.def.relu;
    a.absval = absval; a.lower_bound = lb;
    a.w2 = nullptr; a.bias2 = nullptr; a.y2 = nullptr; a.y2_cs = 0; a.cout2 = 0;
    return a;
  }

  // launch one (possibly fused) layer; fuse as in launch_conv_ks
  int run(const LayerW& L, const ConvArgs& a, int fuse) const {
    ProfRec pr{(int)(&L - net->layers.data()), 0, B, a.Din, nullptr, nullptr};
    // profiling is an aid, not part of the data path: a launch whose events cannot be created / recorded is simply not listed
    const bool timed = net->profiling && hipEventCreate(&pr.t0) == hipSuccess && hipEventCreate(&pr.t1) == hipSuccess &&
                       hipEventRecord(pr.t0, s) == hipSuccess;
    int rc = 0;
    if (net->algo != 1 && fuse == 0 && (L.def.cin == 1 || L.def.cout == 1) && (rc = launch_conv_valu(a, s, true)) != 0) {
      if (rc > 0) { pr.mfma = 7; rc = 0; }       // conv_in / deconv_out: LDS-tiled VALU kernel
    } else if (net->algo != 1 && L.w_mfma) {
      rc = launch_conv_ks(a, L.w_mfma, fuse, s, true);
      if (rc > 0) { pr.mfma = 2 + fuse; rc = 0; }
      else if (rc == 0 && fuse == 0) {
        rc = launch_conv_mfma(a, L.w_mfma, s, true);
        if (rc > 0) { pr.mfma = 1; rc = 0; } else if (rc == 0) rc = launch_conv_direct(a, s);
      } else if (rc == 0) {
        set_error("fused VRN kernel unavailable for a shape it was planned for");
        rc = -1;
      }
    } else {
      if (fuse) { set_error("fused launch requested on the direct path"); return -1; }
      rc = launch_conv_direct(a, s);
    }
    if (timed && hipEventRecord(pr.t1, s) == hipSuccess) net->prof.push_back(pr);
    return rc;
  }

  int conv(const LayerW& L, const float* x, int Din, int x_cs, int x_co, float* y, int y_cs, int y_co,
           const float* res, int absval = 0, float lb = 0.f, int x_q4 = 0, int y_q4 = 0) const {
    ConvArgs a = args(L, x, Din, x_cs, x_co, y, y_cs, y_co, res, absval, lb);
    a.x_q4 = x_q4; a.y_q4 = y_q4;
    return run(L, a, 0);
  }

  // one launch of a row kernel (vrn_row.hip), bracketed by profiling events like run()
  template <class F>
  int row(int layer, int code, int D, F&& launch) const {
    ProfRec pr{layer, code, B, D, nullptr, nullptr};
    const bool timed = net->profiling && hipEventCreate(&pr.t0) == hipSuccess && hipEventCreate(&pr.t1) == hipSuccess &&
                       hipEventRecord(pr.t0, s) == hipSuccess;
    const int rc = launch();
    if (timed && hipEventRecord(pr.t1, s) == hipSuccess) net->prof.push_back(pr);
    return rc;
  }

  // _VoxceptionResNet.call (model_voxception.py:56-68); l = index of conv1_1. x -> out, both [B,D^3,C].
  // q4: x / out are Q4 tensors (the 64^3 stage of the transforms): the row kernels of vrn_row.hip
  // x_nonneg: x is the output of a ReLU (the layer before the stage, or the previous block)
  int vrn(int l, const float* x, float* out, int D, int C, float* t1, float* t2, float* t3, bool q4 = false, bool x_nonneg = false,
          const RowSkip* skipA = nullptr, const RowSkip* skipBC = nullptr) const {
    const auto& Ls = net->layers;
    const int q = C / 4, h = C / 2;
    int rc;
    if (q4) {
      const bool big = C == 16 && D == 64, mid = C == 32 && D == 32, low = C == 64 && D == 16;
      if (!big && !mid && !low) { set_error("Q4 VRN block needs C=16 at D=64, C=32 at D=32 or C=64 at D=16 (got C=%d D=%d)", C, D); return -1; }
      const float* w[10];
      for (int i = 0; i < 5; ++i) { w[2 * i] = Ls[l + i].w_tf; w[2 * i + 1] = Ls[l + i].bias; }
      for (int which = 0; which < (low ? 3 : 2); ++which)       // C = 64: A, B, C (vrn_row16.hip); else A, BC
        if ((rc = row(l + which, low ? (which == 0 ? 8 : 11 + which) : 8 + which, D, [&] {
               return big ? launch_vrn16_row(x, t1, out, w, B, which, s, x_nonneg, which == 0 ? skipA : skipBC)
                          : (mid ? launch_vrn32_row(x, t1, out, w, B, which, s, x_nonneg, which == 0 ? skipA : skipBC, Ls[l].w_row)
                                 : launch_vrn64_row(x, t1, out, w, B, which, s, Ls[l].w_row)); })))
          return rc;
      return 0;
    }
    if (net->algo != 1 && C == 16 && D % 16 == 0) {
      // full-resolution blocks: two VALU kernels (vrn_valu.hip)
      const float* w[10];
      for (int i = 0; i < 5; ++i) { w[2 * i] = Ls[l + i].w_tf; w[2 * i + 1] = Ls[l + i].bias; }
      for (int which = 0; which < 2; ++which) {
        ProfRec pr{l + which, 5 + which, B, D, nullptr, nullptr};
        const bool timed = net->profiling && hipEventCreate(&pr.t0) == hipSuccess && hipEventCreate(&pr.t1) == hipSuccess &&
                           hipEventRecord(pr.t0, s) == hipSuccess;
        rc = launch_vrn16_valu(x, t1, out, w, B, D, which, s);
        if (timed && hipEventRecord(pr.t1, s) == hipSuccess) net->prof.push_back(pr);
        if (rc <= 0) { if (rc == 0) set_error("vrn16 VALU kernel refused D=%d", D); return rc < 0 ? rc : -1; }
      }
      return 0;
    }
    if (net->algo != 1 && Ls[l].w_mfma && Ls[l + 2].w_mfma && Ls[l + 3].w_mfma) {
      // fused form: [conv1_1 + conv2_1] -> conv1_2(+residual) -> [conv2_2 + conv2_3 + residual]
      ConvArgs a1 = args(Ls[l + 0], x, D, C, 0, t1, q, 0, nullptr);
      a1.w2 = Ls[l + 2].w_mfma; a1.bias2 = Ls[l + 2].bias; a1.y2 = t2; a1.y2_cs = q; a1.cout2 = q;
      ConvArgs a3 = args(Ls[l + 3], t2, D, q, 0, out, C, h, x);
      a3.w2 = Ls[l + 4].w_tf; a3.bias2 = Ls[l + 4].bias; a3.cout2 = h;
      if (launch_conv_ks(a1, Ls[l].w_mfma, 1, s, false) == 1 && launch_conv_ks(a3, Ls[l + 3].w_mfma, 2, s, false) == 1) {
        if ((rc = run(Ls[l + 0], a1, 1))) return rc;
        if ((rc = conv(Ls[l + 1], t1, D, q, 0, out, C, 0, x))) return rc;
        return run(Ls[l + 3], a3, 2);
      }
    }
    if ((rc = conv(Ls[l + 0], x, D, C, 0, t1, q, 0, nullptr))) return rc;         // tensor1_1
    if ((rc = conv(Ls[l + 2], x, D, C, 0, t2, q, 0, nullptr))) return rc;         // tensor2_1
    if ((rc = conv(Ls[l + 1], t1, D, q, 0, out, C, 0, x))) return rc;             // relu(x[:h] + tensor1_2)
    if ((rc = conv(Ls[l + 3], t2, D, q, 0, t3, q, 0, nullptr))) return rc;        // tensor2_2
    if ((rc = conv(Ls[l + 4], t3, D, q, 0, out, C, h, x))) return rc;             // relu(x[h:] + tensor2_3)
    return 0;
  }
};

// ---------------------------------------------------------------------------------------------------
// Scheduling of a batch.  The three resolutions of the auto-encoder transforms want different chunk
// sizes: at 64^3 a chunk of a few cubes already gives thousands of workgroups and its activations
// (25 MB / cube with the blocks running in place) should stay inside the 256 MiB Infinity Cache; at 16^3 a cube is only 16 workgroups, so
// ~128 cubes are needed to fill 256 CUs.  A "super chunk" of cubes therefore runs stage by stage, the
// stage boundaries (down_k / up_k outputs) being kept for the whole super chunk.
// ---------------------------------------------------------------------------------------------------
struct Chunks { int big, mid, small; };
// RowSkip scratch per cube of a 64^3 chunk: 64 row-occupancy words (128 floats) + kSkipLaunches tile orders of up to 512
// tiles + kSkipLaunches tables of 64 virtual-row words
constexpr size_t kSkipFloatsPerCube = 128 + (size_t)kSkipLaunches * 512 + (size_t)kSkipLaunches * 128 + (size_t)kSkipLaunchesMid * 256 +
                                      kSkipLaunches + kSkipLaunchesMid;
// ... and of the segment form (vrn_seg.hip): 64 x 64 voxel-occupancy words, kSegLaunches slot lists of up to 1024 slots,
// kSegLaunches tables of 256 bytes, a count per launch; per call: the empty-cube responses the segment kernels and down_1 may read
// (conv_in's, tensor1_1 | tensor2_1 and the outputs of the three blocks), copied next to the tensors when the net's own copy is
// out of the window's reach
constexpr size_t kSegFloatsPerCube = 2 * 4096 + (size_t)kSegLaunches * 1024 + (size_t)kSegLaunches * 64 + kSegLaunches;
constexpr size_t kSegEmptyFloats = (size_t)64 * 64 * 64 * (16 + 3 * 8 + 3 * 16);
constexpr size_t kSegWindowPad = 2u << 20;      // bytes: the window starts this far below the chunk's tensors (SegArgs)

// PCGC_SKIP_EMPTY: 0 = compute every tile; 1 = empty tiles are not written at all, readers take the
// empty-cube response for them (only the stage's last launch materialises its empty tiles, for down_1); 2 = every launch
// copies its empty tiles (all tensors complete); 3 = as 1, with the three C = 16 blocks on SLOTS of 8 planes x 2 rows x 16 voxels
// instead of whole-row tiles (vrn_seg.hip) — the default.  Read per call: tests compare the settings in one process.
static int skip_mode() {
  const char* e = getenv("PCGC_SKIP_EMPTY");
  return e ? atoi(e) : 3;
}
static bool skip_requested() { return skip_mode() != 0; }

static Chunks chunk_plan(const pcgc_net* net) {
  Chunks c{8, 64, 256};
  // the analysis' 64^3 stage with empty-space skipping computes about half of its tiles: 16 cubes per launch keep two
  // heavy waves on every SIMD (one wave alone runs at 0.6 of the pair's rate; measured 8 / 12 / 16 / 24: DESIGN.md §3)
  if (net->kind == PCGC_NET_ANALYSIS && net->E_in && skip_requested()) c.big = 16;
  // the blocks on slots compute a fifth to a third of them: about 36 cubes per launch put two waves on every SIMD once
  // (1 024 slots per cube, four per wave, 2 048 wave places; measured 16 / 32 / 40: profiles/r06_vB_seg_chunks.txt)
  if (net->kind == PCGC_NET_ANALYSIS && net->E_in && skip_mode() == 3) c.big = 40;
  const char* env = getenv(net->kind == PCGC_NET_ANALYSIS ? "PCGC_CHUNKS_A" : "PCGC_CHUNKS_S");   // experiment knobs
  if (!env) env = getenv("PCGC_CHUNKS");            // "big,mid,small" cubes per launch at D, D/2, D/4
  if (env) {
    int a = 0, b = 0, d = 0;
    if (sscanf(env, "%d,%d,%d", &a, &b, &d) == 3 && a > 0 && b > 0 && d > 0) c = Chunks{a, b, d};
  }
  if (net->chunk > 0) c = Chunks{net->chunk, net->chunk, net->chunk};
  if (net->kind == PCGC_NET_ANALYSIS && net->E_in && skip_mode() == 3 && c.big > kSegMaxChunk) c.big = kSegMaxChunk;   // the slot lists' limit
  return c;
}
static inline int imax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
static inline int imin(int a, int b) { return a < b ? a : b; }

// floats of scratch for B cubes (D = input spatial size)
static size_t ws_floats(const pcgc_net* net, int B, int D) {
  const size_t d3 = (size_t)D * D * D;
  switch (net->kind) {
    case PCGC_NET_ANALYSIS:
    case PCGC_NET_SYNTHESIS: {
      const bool ana = net->kind == PCGC_NET_ANALYSIS;
      const size_t V = ana ? d3 : d3 * 64;                                   // voxels at full resolution
      const Chunks c = chunk_plan(net);
      const size_t SC = (size_t)imin(B, imax3(c.big, c.mid, c.small));
      const size_t s2 = SC * (V / 8) * 32, s3 = SC * (ana ? (V / 64) * 64 : 0);
      const size_t wb = (size_t)imin(B, c.big) * V * 16, wm = (size_t)imin(B, c.mid) * (V / 8) * 32,
                   wsm = (size_t)imin(B, c.small) * (V / 64) * 64;
      size_t work = wb > wm ? wb : wm;
      if (wsm > work) work = wsm;
      // one activation tensor (blocks run in place) + VRN scratch + the row-occupancy words of a 64^3 chunk (RowSkip)
      return s2 + s3 + work + (work / 4) * 3 + SC * kSkipFloatsPerCube + 64 + (ana && D == 64 && net->E_in ? SC * kSegFloatsPerCube + kSegEmptyFloats + 256 : 0);
    }
    case PCGC_NET_HYPER_ENCODER:
      return (size_t)imin(B, 256) * (d3 * 16 + d3 * 2);
    case PCGC_NET_HYPER_DECODER:
      return (size_t)imin(B, 256) * (d3 * 16 + d3 * 8 * 16 + d3 * 8 * 32);
  }
  return 0;
}

// three VRN blocks starting at layer l, IN PLACE on `a`: every kernel that writes the block output reads the block
// input only for the residual, at the very element it then overwrites (vrn16_bc, the `res` epilogues), and the
// kernels that read the input with a halo (conv1_1 / conv2_1) run before any of those.  One activation tensor
// instead of two keeps a 64^3 chunk's working set (x + t12 = 201 MB for 8 cubes) inside the 256 MiB Infinity
// Cache, where the ping-pong pair (250 MB at 6 cubes) thrashed it (measured: vrn16_bc 14.6 -> 13.0 ms per step).
static int vrn3(const Exec& E, int l, float* a, int d, int c, float* t, size_t full, float** result, bool q4 = false,
                const unsigned* order = nullptr, const unsigned* n_heavy = nullptr, int cap = 0, const unsigned long long* virt = nullptr,
                const float* const* e_t = nullptr, const float* const* e_o = nullptr, const float* e_first = nullptr,
                bool unread_ok = false) {
  for (int i = 0; i < 3; ++i) {
    // block 0 follows layer l - 1 (conv_in / down_* / deconv_in / up_*: ReLU per the layer table), the others a block
    const bool nonneg = i > 0 || (l > 0 && E.net->layers[l - 1].def.relu);
    // order (analysis; RowSkip): `order` / `n_heavy` start at the stage's first block launch: block i's kernel A is
    // launch 2i, its BC launch 2i + 1; e_t / e_o = the blocks' empty-cube responses, e_first = the stage input's
    RowSkip ka, kbc;
    if (order) {
      ka.counter = kbc.counter = E.net->skip_counter;
      ka.order = order + (size_t)(2 * i) * cap; ka.n_heavy = n_heavy + 2 * i; ka.empty = e_t[i];
      kbc.order = order + (size_t)(2 * i + 1) * cap; kbc.n_heavy = n_heavy + 2 * i + 1; kbc.empty = e_o[i];
      if (virt) {
        // virtual tiles (64^3 stage): `virt` starts at the table of the launch that made the stage input (conv_in), block
        // i's A is table 1 + 2i, its BC table 2 + 2i, B * 64 words each.  A reads the block input (conv_in or the previous
        // block), BC reads A's output with its halo and the block input as residual; the stage's last launch writes
        // all its tiles (down_1 reads them without a table)
        const size_t tb = (size_t)E.B * 64;
        const float* e_prev = i == 0 ? e_first : e_o[i - 1];
        ka.materialize = 0; ka.in_virtual = virt + (size_t)(2 * i) * tb; ka.in_empty = e_prev;
        // the stage's last launch copies its empty tiles for down_1 — all of them, or (down_1 skipping its own empty tiles:
        // `unread_ok`) only those a computed down_1 tile reads
        kbc.materialize = i == 2 ? (unread_ok ? 2 : 1) : 0;
        kbc.in_virtual = virt + (size_t)(1 + 2 * i) * tb; kbc.in_empty = e_t[i];
        kbc.res_virtual = virt + (size_t)(2 * i) * tb; kbc.res_empty = e_prev;
      }
    }
    int rc = E.vrn(l + 5 * i, a, a, d, c, t, t + full / 4, t + full / 2, q4, nonneg, order ? &ka : nullptr, order ? &kbc : nullptr);
    if (rc) return rc;
  }
  *result = a;
  return 0;
}

// The three C = 16 blocks of the analysis' 64^3 stage on slots (vrn_seg.hip), in place on `a`: per block kernel A reads the block
// input (conv_in's output or the previous block's, slots not written there = the producer's empty-cube response) and writes
// tensor1_1 | tensor2_1 for its heavy slots, kernel BC reads those with their halo and the block input as residual.  Launch c of
// the chunk's lists: block i's A = 1 + 2i, BC = 2 + 2i, 0 = conv_in.  down_1 reads the stage's output through the
// last launch's table (SegRead).
struct SegChunk {
  const char* win;                    // window base
  const unsigned* slots;              // this chunk's lists: launch c at + c * n * 1024
  const unsigned* counts;             // heavy slots per launch
  const unsigned char* virt;          // launch c at + c * n * 256
  const float* e_in;                  // the empty-cube responses inside the window: conv_in's, then e_t[3], e_o[3]
  const float* e_t[3];
  const float* e_o[3];
};
static int vrn3_seg(const Exec& E, int l0, float* a, float* t, const SegChunk& k) {
  const auto& Ls = E.net->layers;
  const int n = E.B;
  auto off = [&](const void* p) { return (unsigned)((const char*)p - k.win); };
  for (int i = 0; i < 3; ++i) {
    const int l = l0 + 5 * i;
    const bool nonneg = i > 0 || (l > 0 && Ls[l - 1].def.relu);
    SegArgs sa;
    sa.win = k.win;
    sa.x_off = off(a); sa.t_off = off(t); sa.out_off = off(a);
    sa.w11 = Ls[l].w_tf; sa.b11 = Ls[l].bias; sa.w12 = Ls[l + 1].w_tf; sa.b12 = Ls[l + 1].bias; sa.w21 = Ls[l + 2].w_tf; sa.b21 = Ls[l + 2].bias;
    sa.w22 = Ls[l + 3].w_tf; sa.b22 = Ls[l + 3].bias; sa.w23 = Ls[l + 4].w_tf; sa.b23 = Ls[l + 4].bias;
    const float* e_prev = i == 0 ? k.e_in : k.e_o[i - 1];
    const unsigned char* v_prev = k.virt + (size_t)(2 * i) * n * 256;
    sa.slots = k.slots + (size_t)(1 + 2 * i) * n * 1024; sa.n_slots = k.counts + (1 + 2 * i);
    sa.in_virt = v_prev; sa.ein_off = off(e_prev);
    int rc = E.row(l, 19, 64, [&] { return launch_vrn16_seg(sa, 0, nonneg, n * 1024, E.s); });
    if (rc) return rc;
    sa.slots = k.slots + (size_t)(2 + 2 * i) * n * 1024; sa.n_slots = k.counts + (2 + 2 * i);
    sa.in_virt = k.virt + (size_t)(1 + 2 * i) * n * 256; sa.ein_off = off(k.e_t[i]);
    sa.res_virt = v_prev; sa.eres_off = off(e_prev);
    rc = E.row(l + 1, 20, 64, [&] { return launch_vrn16_seg(sa, 1, nonneg, n * 1024, E.s); });
    if (rc) return rc;
  }
  return 0;
}

static int forward_autoencoder(const pcgc_net* net, const float* x, float* out, int B, int D, float* ws, hipStream_t s) {
  const bool ana = net->kind == PCGC_NET_ANALYSIS;
  const auto& Ls = net->layers;
  const int Db = ana ? D : 4 * D, Dm = Db / 2, Ds = Db / 4;
  const size_t V = (size_t)Db * Db * Db;
  const Chunks ch = chunk_plan(net);
  const int SC = imin(B, imax3(ch.big, ch.mid, ch.small));
  const size_t s2_cube = (V / 8) * 32, s3_cube = ana ? (V / 64) * 64 : 0;      // synthesis keeps no 64^3 stage buffer
  float* S2 = ws;
  float* S3 = S2 + (size_t)SC * s2_cube;
  float* work = S3 + (size_t)SC * s3_cube;
  // the full-resolution stage runs on the row kernels (vrn_row.hip) with its activations in the Q4 layout
  static const int stages = getenv("PCGC_ROW_STAGES") ? atoi(getenv("PCGC_ROW_STAGES")) : 127;   // experiment knob: bit per stage
  const bool q4 = net->algo != 1 && Db == 64 && (stages & 1);
  const bool q4m = net->algo != 1 && Dm == 32 && (stages & 2);     // the middle stage (C = 32 at 32^3) likewise: vrn_row32.hip
  const bool q4s = net->algo != 1 && Ds == 16 && (stages & 4);     // and the low-resolution stage (C = 64 at 16^3): vrn_row16.hip
  // exact skipping of empty space in the analysis' 64^3 stage (RowSkip): PCGC_SKIP_EMPTY=0 computes every tile
  const bool skip = ana && q4 && skip_requested() && net->E_in != nullptr;
  // RowSkip scratch behind the activation tensor + VRN scratch, all sized per cube of the super chunk: 64 row-occupancy
  // words; per 64^3 launch configuration 64 virtual-row words and 512 tile-order entries; per 32^3 configuration 256 entries;
  // then the heavy-tile counts per (chunk, configuration).  Filled once per super chunk (three launches), before the stages.
  unsigned long long* rowocc = nullptr;
  unsigned *order = nullptr, *n_heavy = nullptr, *order_mid = nullptr, *n_heavy_mid = nullptr;
  unsigned long long* virt = nullptr;
  if (skip) {
    const size_t wb = (size_t)imin(B, ch.big) * V * 16, wm = (size_t)imin(B, ch.mid) * s2_cube, wsm = (size_t)imin(B, ch.small) * (V / 64) * 64;
    size_t wk = wb > wm ? wb : wm;
    if (wsm > wk) wk = wsm;
    float* sk = work + wk + (wk / 4) * 3;
    sk += (16 - ((uintptr_t)sk / 4) % 16) % 16;                // 64-byte aligned
    rowocc = reinterpret_cast<unsigned long long*>(sk);
    virt = rowocc + (size_t)SC * 64;
    order = reinterpret_cast<unsigned*>(virt + (size_t)SC * kSkipLaunches * 64);
    order_mid = order + (size_t)SC * kSkipLaunches * 512;
    n_heavy = order_mid + (size_t)SC * kSkipLaunchesMid * 256;
    n_heavy_mid = n_heavy + (size_t)SC * kSkipLaunches;        // (at most one chunk per cube)
  }
  // the blocks on slots (PCGC_SKIP_EMPTY=3): their scratch follows; the empty-cube responses they may read are copied behind
  // it, so that one buffer window of < 2 GiB holds the chunk's tensors and the responses
  const bool seg = skip && skip_mode() == 3 && q4m && (stages & 16) && Ls[16].w_row;     // (down_1's row kernel reads the slot-wise output)
  unsigned long long* occ64 = nullptr;
  unsigned *seg_slots = nullptr, *seg_counts = nullptr;
  unsigned char* seg_virt = nullptr;
  SegChunk segk{};
  if (seg) {
    float* sg = reinterpret_cast<float*>(n_heavy_mid + (size_t)SC * kSkipLaunchesMid);
    sg += (16 - ((uintptr_t)sg / 4) % 16) % 16;
    occ64 = reinterpret_cast<unsigned long long*>(sg);
    seg_slots = reinterpret_cast<unsigned*>(occ64 + (size_t)SC * 4096);
    seg_counts = seg_slots + (size_t)SC * kSegLaunches * 1024;
    seg_virt = reinterpret_cast<unsigned char*>(seg_counts + (size_t)SC * kSegLaunches);
    float* ec = reinterpret_cast<float*>(seg_virt + (size_t)SC * kSegLaunches * 256);
    ec += (64 - ((uintptr_t)ec / 4) % 64) % 64;
    // E_in, E_t[0..2], E_o[0..2] are contiguous in the net's blob (make_empty_responses).  Where the blob itself lies within
    // reach of the chunk's tensors — one window of < 2 GiB covers both — the kernels read it in place; else (or with
    // PCGC_SEG_COPY_EMPTY=1) it is copied behind the scratch, 92 MB per call
    const char* lo = reinterpret_cast<const char*>(work);
    const char* hi = reinterpret_cast<const char*>(ec);          // everything of the chunk lies below the copy's place
    const char* blob_lo = reinterpret_cast<const char*>(net->E_in);
    const char* blob_hi = blob_lo + kSegEmptyFloats * sizeof(float);
    const char* fc = getenv("PCGC_SEG_COPY_EMPTY");                // read per call, like PCGC_SKIP_EMPTY: tests compare both ways
    const bool force_copy = fc && atoi(fc) != 0;
    const char* wlo = blob_lo < lo ? blob_lo : lo;
    const char* whi = blob_hi > hi ? blob_hi : hi;
    if (!force_copy && (size_t)(whi - wlo) + kSegWindowPad < 0x7ffff000u) {
      segk.win = wlo - kSegWindowPad;
      ec = const_cast<float*>(net->E_in);
    } else {
      PCGC_CHECK_HIP(hipMemcpyAsync(ec, net->E_in, kSegEmptyFloats * sizeof(float), hipMemcpyDeviceToDevice, s));
      segk.win = lo - kSegWindowPad;
    }
    segk.e_in = ec;
    for (int i = 0; i < 3; ++i) segk.e_t[i] = ec + V * 16 + (size_t)i * V * 8;
    for (int i = 0; i < 3; ++i) segk.e_o[i] = ec + V * 40 + (size_t)i * V * 16;
    PCGC_REQUIRE((size_t)(reinterpret_cast<const char*>(ec + kSegEmptyFloats) - segk.win) < 0x7ffff000u &&
                 (size_t)(reinterpret_cast<const char*>(seg_virt) - segk.win) < 0x7ffff000u,
                 "analysis: the 64^3 chunk and the empty-cube responses do not fit one 2 GiB buffer window");
  }
  const bool virtual_tiles = skip && (skip_mode() == 1 || skip_mode() == 3);
  // ... and in down_1 + the 32^3 stage (copy mode: every tile stays materialised); PCGC_SKIP_MID=0 stops at the 64^3 stage
  const char* mid_env = getenv("PCGC_SKIP_MID");
  const bool skip_mid = skip && q4m && (stages & 16) && Ls[16].w_row && net->E_d1 && !(mid_env && atoi(mid_env) == 0);
  int rc;
  for (int b0 = 0; b0 < B; b0 += SC) {
    const int nb = imin(SC, B - b0);
    // 64^3 chunks of equal size (103 cubes: 7 x 15 or 14 instead of 6 x 16 + 7): with empty-space skipping a launch takes
    // as long as its fullest SIMD, so a short last chunk costs as much as a full one, and chunks just over the size that
    // fills every wave slot once pay a second round (profiles/r04_vC_skip_launches.txt)
    const int big = (nb + (nb + ch.big - 1) / ch.big - 1) / ((nb + ch.big - 1) / ch.big);
    if (ana) {
      if (skip) {                                              // row occupancy and every chunk's tile orders: they depend on the input only
        if (seg) {
          if ((rc = launch_voxocc(x + (size_t)b0 * V, occ64, rowocc, nb, s))) return rc;
          if ((rc = launch_seg_order(occ64, rowocc, nb, big, seg_slots, seg_counts, seg_virt, net->skip_counter, s))) return rc;
        } else if ((rc = launch_rowocc(x + (size_t)b0 * V, rowocc, nb, s))) return rc;
        if ((rc = launch_tile_order(rowocc, nb, big, net->skip_cfg, nullptr, kSkipLaunches, order, n_heavy, 512, virt, s))) return rc;
        if (skip_mid && (rc = launch_tile_order(rowocc, nb, ch.mid, net->skip_cfg_mid[0], net->skip_cfg_mid[1], kSkipLaunchesMid, order_mid,
                                                n_heavy_mid, 256, nullptr, s))) return rc;
      }
      // 64^3: conv_in, vrn1_*, down_1 -> S2
      for (int c0 = 0; c0 < nb; c0 += big) {
        const int n = imin(big, nb - c0);
        Exec E{net, s, n};
        const size_t full = (size_t)n * V * 16;
        float* A = work; float* t = A + full; float* r;
        const float* xin = x + (size_t)(b0 + c0) * V;
        // this chunk's tile orders (configuration c at + c * cap), counts and virtual-row tables
        const int cap = n * 512;
        const unsigned* ord = skip ? order + (size_t)c0 * kSkipLaunches * 512 : nullptr;
        const unsigned* nhv = skip ? n_heavy + (size_t)(c0 / big) * kSkipLaunches : nullptr;
        const unsigned long long* vrt = skip ? virt + (size_t)c0 * kSkipLaunches * 64 : nullptr;
        RowSkip kin;
        if (skip) {
          kin.order = ord; kin.n_heavy = nhv; kin.empty = net->E_in; kin.counter = net->skip_counter;
          kin.materialize = virtual_tiles ? 0 : 1;
        }
        if (seg) {                                             // conv_in on slots, launch 0 of the chunk's lists
          ConvInSegArgs ca;
          ca.x = xin; ca.win = segk.win; ca.out_off = (unsigned)((const char*)A - segk.win);
          ca.slots = seg_slots + (size_t)c0 * kSegLaunches * 1024; ca.n_slots = seg_counts + (size_t)(c0 / big) * kSegLaunches;
          ca.w = Ls[0].w_tf; ca.bias = Ls[0].bias; ca.relu = Ls[0].def.relu;
          rc = E.row(0, 21, Db, [&] { return launch_conv_in_seg(ca, n * 1024, s); });
        } else if (q4) rc = E.row(0, 10, Db, [&] { return launch_conv_in_row(xin, A, Ls[0].w_tf, Ls[0].bias, n, Ls[0].def.relu, s, skip ? &kin : nullptr); });
        else rc = E.conv(Ls[0], xin, Db, 1, 0, A, 16, 0, nullptr);
        if (rc) return rc;
        if (seg) {
          SegChunk k = segk;
          k.slots = seg_slots + (size_t)c0 * kSegLaunches * 1024;
          k.counts = seg_counts + (size_t)(c0 / big) * kSegLaunches;
          k.virt = seg_virt + (size_t)c0 * kSegLaunches * 256;
          if ((rc = vrn3_seg(E, 1, A, t, k))) return rc;
          r = A;
        } else if ((rc = vrn3(E, 1, A, Db, 16, t, full, &r, q4, skip ? ord + cap : nullptr, skip ? nhv + 1 : nullptr, cap, virtual_tiles ? vrt : nullptr,
                       net->E_t, net->E_o, net->E_in, skip_mid))) return rc;
        float* down_out = S2 + (size_t)c0 * s2_cube;
        RowSkip kd1;                                           // down_1: launch 7 of the chunk's tile orders, copy mode
        if (skip_mid) { kd1.order = ord + (size_t)7 * cap; kd1.n_heavy = nhv + 7; kd1.empty = net->E_d1; kd1.counter = net->skip_counter; }
        SegRead sr;                                            // the blocks ran on slots: down_1 reads their output through the last launch's table
        if (seg) {
          sr.win = segk.win; sr.x_off = (unsigned)((const char*)r - segk.win); sr.e_off = (unsigned)((const char*)segk.e_o[2] - segk.win);
          sr.virt = seg_virt + (size_t)c0 * kSegLaunches * 256 + (size_t)6 * n * 256;
        }
        if (q4 && q4m && (stages & 16) && Ls[16].w_row) rc = E.row(16, 15, Db, [&] { return launch_down1_row(r, down_out, Ls[16].w_row, Ls[16].bias, n, Ls[16].def.relu, s, skip_mid ? &kd1 : nullptr, false, nullptr, seg ? &sr : nullptr); });
        else if (seg) { set_error("analysis: the segment form of the 64^3 blocks needs down_1's row kernel"); return -1; }
        else rc = E.conv(Ls[16], r, Db, 16, 0, down_out, 32, 0, nullptr, 0, 0.f, q4, q4m);
        if (rc) return rc;
      }
      // 32^3: vrn2_*, down_2 -> S3
      for (int c0 = 0; c0 < nb; c0 += ch.mid) {
        const int n = imin(ch.mid, nb - c0);
        Exec E{net, s, n};
        const size_t full = (size_t)n * s2_cube;
        float* t = work; float* r;                    // the blocks run in place on the stage buffer
        // this chunk's six tile orders (made for the tiles launch_vrn32_row uses at this launch size)
        const unsigned* ordm = skip_mid ? order_mid + (size_t)c0 * kSkipLaunchesMid * 256 : nullptr;
        const unsigned* nhm = skip_mid ? n_heavy_mid + (size_t)(c0 / ch.mid) * kSkipLaunchesMid : nullptr;
        if ((rc = vrn3(E, 17, S2 + (size_t)c0 * s2_cube, Dm, 32, t, full, &r, q4m, ordm, nhm, n * 256, nullptr, net->E_t32, net->E_o32, net->E_d1))) return rc;
        float* down2_out = S3 + (size_t)c0 * s3_cube;
        if (q4m && q4s && (stages & 64) && Ls[32].w_row) rc = E.row(32, 15, Dm, [&] { return launch_down2_row(r, down2_out, Ls[32].w_row, Ls[32].bias, n, Ls[32].def.relu, s); });
        else rc = E.conv(Ls[32], r, Dm, 32, 0, down2_out, 64, 0, nullptr, 0, 0.f, q4m, q4s);
        if (rc) return rc;
      }
      // 16^3: vrn3_*, conv_out
      for (int c0 = 0; c0 < nb; c0 += ch.small) {
        const int n = imin(ch.small, nb - c0);
        Exec E{net, s, n};
        const size_t full = (size_t)n * s3_cube;
        float* t = work; float* r;                    // the blocks run in place on the stage buffer
        if ((rc = vrn3(E, 33, S3 + (size_t)c0 * s3_cube, Ds, 64, t, full, &r, q4s))) return rc;
        if ((rc = E.conv(Ls[48], r, Ds, 64, 0, out + (size_t)(b0 + c0) * (V / 64) * 16, 16, 0, nullptr, 0, 0.f, q4s, 0))) return rc;
      }
    } else {
      // 16^3: deconv_in, vrn1_*, up_1 -> S2
      for (int c0 = 0; c0 < nb; c0 += ch.small) {
        const int n = imin(ch.small, nb - c0);
        Exec E{net, s, n};
        const size_t full = (size_t)n * (V / 64) * 64;
        float* A = work; float* t = A + full; float* r;
        if ((rc = E.conv(Ls[0], x + (size_t)(b0 + c0) * (V / 64) * 16, Ds, 16, 0, A, 64, 0, nullptr, 0, 0.f, 0, q4s))) return rc;
        if ((rc = vrn3(E, 1, A, Ds, 64, t, full, &r, q4s))) return rc;
        float* up1_out = S2 + (size_t)c0 * s2_cube;
        if (q4s && q4m && (stages & 32) && Ls[16].w_row) rc = E.row(16, 14, Ds, [&] { return launch_up1_row(r, up1_out, Ls[16].w_row, Ls[16].bias, n, Ls[16].def.relu, s); });
        else rc = E.conv(Ls[16], r, Ds, 64, 0, up1_out, 32, 0, nullptr, 0, 0.f, q4s, q4m);
        if (rc) return rc;
      }
      // 32^3: vrn2_* in place on S2
      for (int c0 = 0; c0 < nb; c0 += ch.mid) {
        const int n = imin(ch.mid, nb - c0);
        Exec E{net, s, n};
        const size_t full = (size_t)n * s2_cube;
        float* t = work; float* r;
        if ((rc = vrn3(E, 17, S2 + (size_t)c0 * s2_cube, Dm, 32, t, full, &r, q4m))) return rc;
      }
      // 64^3: up_2, vrn3_*, deconv_out per chunk — the 16-channel full-resolution tensor (16.8 MB per cube) never
      // makes the round trip through HBM: up_2 writes it chunk by chunk right before the blocks that consume it
      for (int c0 = 0; c0 < nb; c0 += big) {
        const int n = imin(big, nb - c0);
        Exec E{net, s, n};
        const size_t full = (size_t)n * V * 16;
        float* A = work; float* t = A + full; float* r;
        const float* up_in = S2 + (size_t)c0 * s2_cube;
        if (q4 && q4m && (stages & 8) && Ls[32].w_row) rc = E.row(32, 14, Dm, [&] { return launch_up2_row(up_in, A, Ls[32].w_row, Ls[32].bias, n, Ls[32].def.relu, s); });
        else rc = E.conv(Ls[32], up_in, Dm, 32, 0, A, 16, 0, nullptr, 0, 0.f, q4m, q4);
        if (rc) return rc;
        if ((rc = vrn3(E, 33, A, Db, 16, t, full, &r, q4))) return rc;
        float* yout = out + (size_t)(b0 + c0) * V;
        if (q4) rc = E.row(48, 11, Db, [&] { return launch_deconv_out_row(r, yout, Ls[48].w_tf, Ls[48].bias, n, Ls[48].def.relu, s); });
        else rc = E.conv(Ls[48], r, Db, 16, 0, yout, 1, 0, nullptr);
        if (rc) return rc;
      }
    }
  }
  return 0;
}

static int forward_chunk(const pcgc_net* net, const float* x, float* out0, float* out1, int B, int D,
                         float lb, float* ws, hipStream_t s) {
  Exec E{net, s, B};
  const auto& Ls = net->layers;
  int rc;
  if (net->kind == PCGC_NET_HYPER_ENCODER) {
    const size_t d3 = (size_t)B * D * D * D;
    float* f1 = ws;
    float* f2 = f1 + d3 * 16;
    if ((rc = E.conv(Ls[0], x, D, 16, 0, f1, 16, 0, nullptr))) return rc;
    if (net->algo != 1 && D == 16) rc = E.row(1, 18, 16, [&] { return launch_down8_row(f1, f2, Ls[1].w_tf, Ls[1].bias, B, Ls[1].def.relu, s); });
    else rc = E.conv(Ls[1], f1, D, 16, 0, f2, 16, 0, nullptr);
    if (rc) return rc;
    if (net->algo != 1 && D / 2 == 8) {                   // 8^3: plane-vector row kernel (hyper_row.hip)
      rc = E.row(2, 16, 8, [&] { return launch_conv8_row(f2, out0, Ls[2].w_tf, Ls[2].bias, B, 16, 8, Ls[2].def.relu, s); });
      if (rc != 0) return rc < 0 ? rc : 0;
    }
    return E.conv(Ls[2], f2, D / 2, 16, 0, out0, 8, 0, nullptr);
  }
  if (net->kind == PCGC_NET_HYPER_DECODER) {
    const size_t d3 = (size_t)B * D * D * D;
    float* f1 = ws;
    float* f2 = f1 + d3 * 16;
    float* f3 = f2 + d3 * 8 * 16;
    if (net->algo != 1 && D == 8) {                       // 8^3 layers: plane-vector row kernels (hyper_row.hip)
      rc = E.row(0, 16, 8, [&] { return launch_conv8_row(x, f1, Ls[0].w_tf, Ls[0].bias, B, 8, 16, Ls[0].def.relu, s); });
      if (rc < 0) return rc;
      if (rc == 0) { set_error("hyper decoder conv1: no 8^3 row kernel for 8 -> 16"); return -1; }
      if ((rc = E.row(1, 17, 8, [&] { return launch_up8_row(f1, f2, Ls[1].w_tf, Ls[1].bias, B, Ls[1].def.relu, s); }))) return rc;
    } else {
      if ((rc = E.conv(Ls[0], x, D, 8, 0, f1, 16, 0, nullptr))) return rc;
      if ((rc = E.conv(Ls[1], f1, D, 16, 0, f2, 16, 0, nullptr))) return rc;
    }
    if ((rc = E.conv(Ls[2], f2, 2 * D, 16, 0, f3, 32, 0, nullptr))) return rc;
    if ((rc = E.conv(Ls[3], f3, 2 * D, 32, 0, out0, 16, 0, nullptr))) return rc;
    return E.conv(Ls[4], f3, 2 * D, 32, 0, out1, 16, 0, nullptr, /*absval=*/1, lb);  // |scale| clamped
  }
  set_error("unknown net kind %d", net->kind);
  return -1;
}

static size_t out_floats_per_cube(int kind, int D, int which) {
  const size_t d3 = (size_t)D * D * D;
  switch (kind) {
    case PCGC_NET_ANALYSIS: return which == 0 ? d3 / 64 * 16 : 0;
    case PCGC_NET_SYNTHESIS: return which == 0 ? d3 * 64 : 0;
    case PCGC_NET_HYPER_ENCODER: return which == 0 ? d3 / 8 * 8 : 0;
    case PCGC_NET_HYPER_DECODER: return d3 * 8 * 16;
  }
  return 0;
}
static size_t in_floats_per_cube(int kind, int D) {
  const size_t d3 = (size_t)D * D * D;
  switch (kind) {
    case PCGC_NET_ANALYSIS: return d3;
    case PCGC_NET_SYNTHESIS: return d3 * 16;
    case PCGC_NET_HYPER_ENCODER: return d3 * 16;
    case PCGC_NET_HYPER_DECODER: return d3 * 8;
  }
  return 0;
}

}  // namespace pcgc

using namespace pcgc;

// The analysis' 64^3 stage applied to ONE all-zero cube, kept per layer output (RowSkip): the very kernels of the forward
// pass, every tile computed — so a skipped tile's copy is bit-identical to what the wave would have computed.
static int make_empty_responses(pcgc_net* net, hipStream_t s) {
  const size_t V = 64 * 64 * 64, Vm = 32 * 32 * 32;
  const size_t n64 = V * (1 + 16 + 3 * 8 + 3 * 16), n32 = Vm * (32 + 3 * 16 + 3 * 32);
  const size_t total = n64 + n32 + 256;
  float* b = nullptr;
  PCGC_CHECK_HIP(hipMalloc(&b, total * sizeof(float)));
  net->empty_blob = b;
  // launch geometries (tile rows x planes: vrn_row.hip, vrn_row32.hip) and the fine (64^3) window each output depends on.
  // 64^3 stage: conv_in radius 1; block i: tensor1_1 2 + 2i (tensor2_1 less: it shares the launch), block output 3 + 2i.
  // down_1 (stride 2, nothing padded in front, one voxel behind): output o reads fine 2o .. 2o + 2 of a radius-7 tensor
  // = [2o - 7, 2o + 9]; every 3^3 layer at 32^3 adds two fine voxels on each side.
  TileCfg cfg[kSkipLaunches + 2 * kSkipLaunchesMid] = {
      {2, 4, 1, 1, 1}, {2, 8, 2, 2, 1}, {2, 8, 3, 3, 1}, {2, 8, 4, 4, 1}, {2, 8, 5, 5, 1}, {2, 8, 6, 6, 1},
      // the stage's last launch: a down_1 tile (2 x 2 outputs at 32^3) reads fine rows / planes [2o, 2o + 4] and is computed
      // iff [2o - 7, 2o + 11] holds an occupied row, so an empty tile here is read only if its own rows / planes dilated by
      // 4 + 7 = 11 do (TileCfg::need)
      {2, 8, 7, 7, 1, 4 + 7},
      {kDown1TileRows, kDown1TilePlanes, 7, 9, 2}};
  static_assert(kDown1TileRows == 2 && kDown1TilePlanes == 2, "the `need` radius above is derived for 2 x 2 down_1 tiles");
  for (int v = 0; v < 2; ++v)                                  // v = 0: launches of > 16 cubes, v = 1: small launches
    for (int i = 0; i < kSkipLaunchesMid; ++i) {
      int th, ld;
      vrn32_tile_geometry(v == 0 ? 64 : 1, i & 1, &th, &ld);
      cfg[kSkipLaunches + v * kSkipLaunchesMid + i] = TileCfg{th, ld, 9 + 2 * i, 11 + 2 * i, 2};
    }
  TileCfg* cfg_dev = reinterpret_cast<TileCfg*>(b + n64 + n32);
  static_assert(sizeof(cfg) <= 256 * sizeof(float), "the configuration tables fit behind the tensors");
  PCGC_CHECK_HIP(hipMemcpyAsync(cfg_dev, cfg, sizeof(cfg), hipMemcpyHostToDevice, s));
  PCGC_CHECK_HIP(hipStreamSynchronize(s));                   // cfg lives on this stack frame
  net->skip_cfg = cfg_dev;
  net->skip_cfg_mid[0] = cfg_dev + kSkipLaunches;
  net->skip_cfg_mid[1] = cfg_dev + kSkipLaunches + kSkipLaunchesMid;
  float* zero = b;
  float* e_in = zero + V;
  float* e_t[3];
  float* e_o[3];
  float* p = e_in + V * 16;
  for (int i = 0; i < 3; ++i) { e_t[i] = p; p += V * 8; }
  for (int i = 0; i < 3; ++i) { e_o[i] = p; p += V * 16; }
  PCGC_CHECK_HIP(hipMemsetAsync(zero, 0, V * sizeof(float), s));
  const auto& Ls = net->layers;
  int rc = launch_conv_in_row(zero, e_in, Ls[0].w_tf, Ls[0].bias, 1, Ls[0].def.relu, s);
  const float* x = e_in;
  for (int i = 0; i < 3 && !rc; ++i) {
    const int l = 1 + 5 * i;
    const float* w[10];
    for (int k = 0; k < 5; ++k) { w[2 * k] = Ls[l + k].w_tf; w[2 * k + 1] = Ls[l + k].bias; }
    rc = launch_vrn16_row(x, e_t[i], e_o[i], w, 1, 0, s, true);
    if (!rc) rc = launch_vrn16_row(x, e_t[i], e_o[i], w, 1, 1, s, true);      // x_nonneg as in the forward pass (bit-identical either way)
    x = e_o[i];
  }
  if (rc) return rc;
  net->E_in = e_in;
  for (int i = 0; i < 3; ++i) { net->E_t[i] = e_t[i]; net->E_o[i] = e_o[i]; }
  // 32^3: down_1 and the three C = 32 blocks (the kernels' sums do not depend on the tile variant a launch size picks)
  if (Ls[16].w_row) {
    float* e_d1 = p; p += Vm * 32;
    float* e_t32[3];
    float* e_o32[3];
    for (int i = 0; i < 3; ++i) { e_t32[i] = p; p += Vm * 16; }
    for (int i = 0; i < 3; ++i) { e_o32[i] = p; p += Vm * 32; }
    rc = launch_down1_row(e_o[2], e_d1, Ls[16].w_row, Ls[16].bias, 1, Ls[16].def.relu, s);
    const float* xm = e_d1;
    for (int i = 0; i < 3 && !rc; ++i) {
      const int l = 17 + 5 * i;
      const float* w[10];
      for (int k = 0; k < 5; ++k) { w[2 * k] = Ls[l + k].w_tf; w[2 * k + 1] = Ls[l + k].bias; }
      rc = launch_vrn32_row(xm, e_t32[i], e_o32[i], w, 1, 0, s, true);
      if (!rc) rc = launch_vrn32_row(xm, e_t32[i], e_o32[i], w, 1, 1, s, true);
      xm = e_o32[i];
    }
    if (rc) return rc;
    net->E_d1 = e_d1;
    for (int i = 0; i < 3; ++i) { net->E_t32[i] = e_t32[i]; net->E_o32[i] = e_o32[i]; }
  }
  return 0;
}

extern "C" {

int pcgc_version(void) { return 1; }
const char* pcgc_last_error(void) { return pcgc::g_err; }

int pcgc_net_param_count(int kind) {
  int n = 0;
  for (const auto& d : layer_table(kind)) n += 1 + d.bias;
  return n;
}

int pcgc_net_create(int kind, const float* const* params, int n_params, pcgc_stream_t stream, pcgc_net** out) {
  hipStream_t s = (hipStream_t)stream;
  auto table = layer_table(kind);
  PCGC_REQUIRE(!table.empty(), "pcgc_net_create: unknown kind %d", kind);
  PCGC_REQUIRE(n_params == pcgc_net_param_count(kind), "pcgc_net_create: kind %d expects %d tensors, got %d", kind,
               pcgc_net_param_count(kind), n_params);
  PCGC_REQUIRE(out != nullptr, "pcgc_net_create: out is NULL");
  size_t total = 0;
  auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };  // 256-byte aligned sub-buffers
  for (const auto& d : table) {
    const size_t wn = (size_t)d.k * d.k * d.k * d.cin * d.cout;
    total += al(wn) + (d.bias ? al(d.cout) : 0);
    total += al(mfma_packed_floats(d.cin, d.cout, d.k, mode_of(d)));
    total += al(row_image_floats(d.cin, d.cout, d.k, mode_of(d)));
    if (d.tconv && d.k == 3 && d.cin == 64 && d.cout == 32) total += al(up1_image_floats());
    if (!d.tconv && d.stride == 2 && d.k == 3 && d.cin == 32 && d.cout == 64) total += al(down2_image_floats());
    if (strcmp(d.name, "conv1_1") == 0 && d.cin == 64 && d.cout == 16) total += al(vrn64_image_floats());      // a C = 64 block's LDS images
    if (strcmp(d.name, "conv1_1") == 0 && d.cin == 32 && d.cout == 8) total += al(vrn32_image_floats());        // a C = 32 block's
  }
  float* blob = nullptr;
  PCGC_CHECK_HIP(hipMalloc(&blob, total * sizeof(float)));
  pcgc_net* net = new pcgc_net();
  net->kind = kind;
  net->algo = 0;
  const char* env = getenv("PCGC_CHUNK_CUBES");
  net->chunk = env ? atoi(env) : 0;
  net->blob = blob;
  float* p = blob;
  int pi = 0;
  for (const auto& d : table) {
    LayerW L;
    L.def = d;
    const size_t wn = (size_t)d.k * d.k * d.k * d.cin * d.cout;
    hipError_t e = hipMemcpyAsync(p, params[pi++], wn * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) { set_error("weight copy failed: %s", hipGetErrorString(e)); pcgc_net_destroy(net); return -100; }
    L.w_tf = p;
    p += al(wn);
    L.bias = nullptr;
    if (d.bias) {
      e = hipMemcpyAsync(p, params[pi++], d.cout * sizeof(float), hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) { set_error("bias copy failed: %s", hipGetErrorString(e)); pcgc_net_destroy(net); return -100; }
      L.bias = p;
      p += al(d.cout);
    }
    L.w_mfma = nullptr;
    if (mfma_packed_floats(d.cin, d.cout, d.k, mode_of(d)) > 0) {
      int rc = pack_weights_mfma(L.w_tf, p, d.cin, d.cout, d.k, mode_of(d), s);
      if (rc) { pcgc_net_destroy(net); return rc; }
      L.w_mfma = p;
      p += al(mfma_packed_floats(d.cin, d.cout, d.k, mode_of(d)));
    }
    if (row_image_floats(d.cin, d.cout, d.k, mode_of(d)) > 0) {
      int rc = launch_row_image(L.w_tf, p, mode_of(d), s);
      if (rc) { pcgc_net_destroy(net); return rc; }
      L.w_row = p;
      p += al(row_image_floats(d.cin, d.cout, d.k, mode_of(d)));
    }
    if (d.tconv && d.k == 3 && d.cin == 64 && d.cout == 32) {          // up_1: image for up1_row_kernel
      int rc = launch_up1_image(L.w_tf, p, s);
      if (rc) { pcgc_net_destroy(net); return rc; }
      L.w_row = p;
      p += al(up1_image_floats());
    }
    if (!d.tconv && d.stride == 2 && d.k == 3 && d.cin == 32 && d.cout == 64) {   // down_2: image for down2_row_kernel
      int rc = launch_down2_image(L.w_tf, p, s);
      if (rc) { pcgc_net_destroy(net); return rc; }
      L.w_row = p;
      p += al(down2_image_floats());
    }
    net->layers.push_back(L);
  }
  // the C = 64 and C = 32 blocks' LDS images (vrn_row16.hip, vrn_row32.hip): conv1_1 + conv2_1 | conv1_2 | conv2_2 of a block in one
  // image, kept with conv1_1
  for (size_t l = 0; l + 4 < net->layers.size(); ++l) {
    const LayerDef& d = net->layers[l].def;
    const bool c64 = d.cin == 64 && d.cout == 16, c32 = d.cin == 32 && d.cout == 8;
    if (strcmp(d.name, "conv1_1") != 0 || !(c64 || c32)) continue;
    const float* w[10];
    for (int i = 0; i < 5; ++i) { w[2 * i] = net->layers[l + i].w_tf; w[2 * i + 1] = net->layers[l + i].bias; }
    int rc = c64 ? launch_vrn64_image(w, p, s) : launch_vrn32_image(w, p, s);
    if (rc) { pcgc_net_destroy(net); return rc; }
    net->layers[l].w_row = p;
    p += al(c64 ? vrn64_image_floats() : vrn32_image_floats());
  }
  if (kind == PCGC_NET_ANALYSIS) {
    int rc = make_empty_responses(net, s);
    if (rc) { pcgc_net_destroy(net); return rc; }
  }
  *out = net;
  return 0;
}

void pcgc_net_destroy(pcgc_net* net) {
  if (!net) return;
  if (net->blob) (void)hipFree(net->blob);
  if (net->empty_blob) (void)hipFree(net->empty_blob);
  delete net;
}

int pcgc_net_set_profiling(pcgc_net* net, int on) {
  PCGC_REQUIRE(net, "pcgc_net_set_profiling: net is NULL");
  net->profiling = on != 0;
  return 0;
}

// One line per launch since the last report: "layer kernel cin cout k mode B Din ms\n".  Synchronises the stream's
// recorded events (profiling aid, not part of the data path).
int pcgc_net_profile_report(pcgc_net* net, char* buf, size_t cap, size_t* needed) {
  PCGC_REQUIRE(net && needed, "pcgc_net_profile_report: NULL argument");
  std::string out;
  for (auto& r : net->prof) {
    float ms = 0.f;
    (void)hipEventSynchronize(r.t1);
    (void)hipEventElapsedTime(&ms, r.t0, r.t1);
    const auto& d = net->layers[r.layer].def;
    char line[256];
    snprintf(line, sizeof(line), "%d %s %s %d %d %d %d %d %d %.6f\n", r.layer, d.name, (r.mfma == 0 ? "direct" : r.mfma == 1 ? "mfma" : r.mfma == 2 ? "ks" : r.mfma == 3 ? "ks1" : r.mfma == 4 ? "ks2" : r.mfma == 5 ? "vrnA" : r.mfma == 6 ? "vrnBC" : r.mfma == 8 ? "rowA" : r.mfma == 9 ? "rowBC" : r.mfma == 10 ? "rowin" : r.mfma == 11 ? "rowout" : r.mfma == 12 ? "rowB" : r.mfma == 13 ? "rowC" : r.mfma == 14 ? "rowup" : r.mfma == 15 ? "rowdown" : r.mfma == 16 ? "rowh8" : r.mfma == 17 ? "rowhup" : r.mfma == 18 ? "rowhdown" : r.mfma == 19 ? "segA" : r.mfma == 20 ? "segBC" : r.mfma == 21 ? "segin" : "valu"), d.cin,
             d.cout, d.k, mode_of(d), r.B, r.Din, ms);
    out += line;
    (void)hipEventDestroy(r.t0);
    (void)hipEventDestroy(r.t1);
  }
  net->prof.clear();
  *needed = out.size() + 1;
  if (buf && cap >= out.size() + 1) memcpy(buf, out.c_str(), out.size() + 1);
  return 0;
}

// Test aid for the exact skipping of empty space (analysis, 64^3 stage): a device word the kernels add 1 to for every
// wave tile they copy from the empty-cube response instead of computing it (NULL: off).
int pcgc_net_set_skip_counter(pcgc_net* net, unsigned* device_counter) {
  PCGC_REQUIRE(net, "pcgc_net_set_skip_counter: net is NULL");
  net->skip_counter = device_counter;
  return 0;
}

int pcgc_rowocc(const float* x, unsigned long long* rowocc, int B, pcgc_stream_t stream) {
  if (B == 0) return 0;
  PCGC_REQUIRE(x && rowocc && B > 0, "pcgc_rowocc: bad arguments");
  return launch_rowocc(x, rowocc, B, (hipStream_t)stream);
}

int pcgc_net_set_algo(pcgc_net* net, int algo) {
  PCGC_REQUIRE(net && (algo == 0 || algo == 1), "pcgc_net_set_algo: bad arguments");
  net->algo = algo;
  return 0;
}

size_t pcgc_net_workspace_bytes(const pcgc_net* net, int B, int D) {
  if (!net || B <= 0) return 0;
  return ws_floats(net, B, D) * sizeof(float) + 256;
}

int pcgc_net_forward(const pcgc_net* net, const float* x, float* out0, float* out1, int B, int D,
                     float scale_lower_bound, void* workspace, size_t workspace_bytes, pcgc_stream_t stream) {
  PCGC_REQUIRE(net != nullptr, "pcgc_net_forward: net is NULL");
  if (B == 0) return 0;                                    // empty batch: valid no-op
  PCGC_REQUIRE(B > 0 && D > 0, "pcgc_net_forward: bad B=%d D=%d", B, D);
  const int div = net->kind == PCGC_NET_ANALYSIS ? 4 : (net->kind == PCGC_NET_HYPER_ENCODER ? 2 : 1);
  PCGC_REQUIRE(D % div == 0, "pcgc_net_forward: input size %d must be a multiple of %d for this transform", D, div);
  PCGC_REQUIRE(x && out0 && (net->kind != PCGC_NET_HYPER_DECODER || out1), "pcgc_net_forward: NULL tensor");
  PCGC_REQUIRE(workspace_bytes >= pcgc_net_workspace_bytes(net, B, D), "pcgc_net_forward: workspace too small (%zu < %zu)",
               workspace_bytes, pcgc_net_workspace_bytes(net, B, D));
  float* ws = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  if (net->kind == PCGC_NET_ANALYSIS || net->kind == PCGC_NET_SYNTHESIS)
    return forward_autoencoder(net, x, out0, B, D, ws, (hipStream_t)stream);
  const int chunk = 256;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = (B - b0) < chunk ? (B - b0) : chunk;
    int rc = forward_chunk(net, x + (size_t)b0 * in_floats_per_cube(net->kind, D),
                           out0 + (size_t)b0 * out_floats_per_cube(net->kind, D, 0),
                           out1 ? out1 + (size_t)b0 * out_floats_per_cube(net->kind, D, 1) : nullptr, nb, D,
                           scale_lower_bound, ws, (hipStream_t)stream);
    if (rc) return rc;
  }
  return 0;
}

// One _VoxceptionResNet block (model_voxception.py:56-68) on NDHWC tensors.  C = 16 at D = 64 runs the row kernels
// of vrn_row.hip (layout conversion in, two kernels, conversion out); other shapes run the generic layer kernels.
size_t pcgc_vrn_workspace_bytes(int B, int D, int C) {
  if (B <= 0 || D <= 0 || C <= 0) return 0;
  const size_t vox = (size_t)B * D * D * D;
  return vox * (size_t)(C + C) * sizeof(float) + 256;      // row path: x in Q4 + t12; generic path: 3 x C/4 scratch
}

int pcgc_vrn_fwd_train_supported(int D, int C) { return (D == 64 && C == 16) || (D == 32 && C == 32); }

int pcgc_vrn_fwd_train(const float* x, const float* const* params, float* t11, float* t21, float* t22, float* pre, float* out, int B,
                       int D, int C, pcgc_stream_t stream) {
  if (B == 0) return 0;
  PCGC_REQUIRE(x && params && t11 && t21 && t22 && pre && out, "pcgc_vrn_fwd_train: NULL tensor");
  PCGC_REQUIRE(pcgc_vrn_fwd_train_supported(D, C), "pcgc_vrn_fwd_train: no fused kernel for D=%d C=%d (run the block layer by layer)", D, C);
  PCGC_REQUIRE(out != x, "pcgc_vrn_fwd_train: the reverse pass needs x, out must not alias it");
  if (D == 32) return launch_vrn32_row_train(x, t11, t21, t22, pre, out, params, B, (hipStream_t)stream);
  return launch_vrn16_row_train(x, t11, t21, t22, pre, out, params, B, (hipStream_t)stream);
}

int pcgc_vrn_fwd_train_signs_supported(int D, int C) { return (D == 64 && C == 16) || (D == 32 && C == 32); }

int pcgc_vrn_fwd_train_signs(const float* x, const float* const* params, float* t11, float* t21, float* t22, int32_t* pre_signs,
                             float* out, int B, int D, int C, pcgc_stream_t stream) {
  if (B == 0) return 0;
  PCGC_REQUIRE(x && params && t11 && t21 && t22 && pre_signs && out, "pcgc_vrn_fwd_train_signs: NULL tensor");
  PCGC_REQUIRE(pcgc_vrn_fwd_train_signs_supported(D, C), "pcgc_vrn_fwd_train_signs: D=%d C=%d (D = 64 with C = 16, D = 32 with C = 32)", D, C);
  PCGC_REQUIRE(out != x, "pcgc_vrn_fwd_train_signs: the reverse pass needs x, out must not alias it");
  if (D == 32) return launch_vrn32_row_train(x, t11, t21, t22, nullptr, out, params, B, (hipStream_t)stream, pre_signs);
  return launch_vrn16_row_train(x, t11, t21, t22, nullptr, out, params, B, (hipStream_t)stream, pre_signs);
}

// the same with x / out in the Q4 layout [b][d][h][C/4][w][4] (the training step's 64^3 stage, Trainer(q4=True))
int pcgc_vrn_fwd_train_q4(const float* x, const float* const* params, float* t11, float* t21, float* t22, int32_t* pre_signs,
                          float* out, int B, int D, int C, pcgc_stream_t stream) {
  if (B == 0) return 0;
  PCGC_REQUIRE(x && params && t11 && t21 && t22 && pre_signs && out, "pcgc_vrn_fwd_train_q4: NULL tensor");
  PCGC_REQUIRE(D == 64 && C == 16, "pcgc_vrn_fwd_train_q4: D=%d C=%d (D = 64 with C = 16 only)", D, C);
  PCGC_REQUIRE(out != x, "pcgc_vrn_fwd_train_q4: the reverse pass needs x, out must not alias it");
  return launch_vrn16_row_train(x, t11, t21, t22, nullptr, out, params, B, (hipStream_t)stream, pre_signs, true);
}

// NDHWC [B][D^3][C] <-> Q4 [B][D][D][C/4][D][4] (to_q4 = 1 / 0); C a multiple of 4
int pcgc_layout_q4(const float* src, float* dst, int B, int D, int C, int to_q4, pcgc_stream_t stream) {
  if (B == 0) return 0;
  PCGC_REQUIRE(src && dst && src != dst && B > 0 && D > 0 && C >= 4 && C % 4 == 0, "pcgc_layout_q4: bad arguments");
  return launch_q4_convert(src, dst, B, D, C, to_q4, (hipStream_t)stream);
}

int pcgc_vrn_fwd(const float* x, const float* const* params, float* out, int B, int D, int C, void* workspace,
                 size_t workspace_bytes, pcgc_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) return 0;
  PCGC_REQUIRE(x && params && out, "pcgc_vrn_fwd: NULL tensor");
  PCGC_REQUIRE(B > 0 && D > 0 && C >= 4 && C % 4 == 0, "pcgc_vrn_fwd: bad B=%d D=%d C=%d", B, D, C);
  PCGC_REQUIRE(workspace && workspace_bytes >= pcgc_vrn_workspace_bytes(B, D, C), "pcgc_vrn_fwd: workspace too small");
  float* ws = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t vox = (size_t)B * D * D * D;
  if ((C == 16 && D == 64) || (C == 32 && D == 32) || (C == 64 && D == 16)) {
    float* xq = ws;
    float* t12 = ws + vox * C;
    int rc;
    if ((rc = launch_q4_convert(x, xq, B, D, C, 1, s))) return rc;
    for (int which = 0; which < (C == 64 ? 3 : 2); ++which)
      if ((rc = C == 16 ? launch_vrn16_row(xq, t12, xq, params, B, which, s)
                        : (C == 32 ? launch_vrn32_row(xq, t12, xq, params, B, which, s) : launch_vrn64_row(xq, t12, xq, params, B, which, s))))
        return rc;
    return launch_q4_convert(xq, out, B, D, C, 0, s);
  }
  pcgc_net net;
  net.kind = -1; net.algo = 1; net.chunk = 0; net.blob = nullptr;
  std::vector<LayerDef> defs;
  push_vrn(defs, C);
  for (int i = 0; i < 5; ++i) net.layers.push_back(LayerW{defs[i], params[2 * i], params[2 * i + 1], nullptr});
  Exec E{&net, s, B};
  const size_t q = vox * (C / 4);
  return E.vrn(0, x, out, D, C, ws, ws + q, ws + 2 * q);
}

#ifdef PCGC_EXPERIMENTS
// experiments (tools/exp/t_ablate.py; builds with PCGC_EXPERIMENTS=1 only — the default libpcgc_hip.so neither exports these
// nor has the global they set): one launch of the 64^3 row kernel A (which = 0) or BC (1) on Q4 tensors already on the device
int pcgc_exp_vrn16_row(const float* xq, float* t12, float* outq, const float* const* params, int B, int which, int x_nonneg, int abl,
                       void* stream) {
  pcgc::g_vrn16_abl = abl;
  const int rc = launch_vrn16_row(xq, t12, outq, params, B, which, (hipStream_t)stream, x_nonneg != 0);
  pcgc::g_vrn16_abl = 0;
  return rc;
}

// the ablation switches for every later launch of the 64^3 row kernels (training variants included)
int pcgc_exp_set_vrn16_ablation(int abl) { pcgc::g_vrn16_abl = abl; return 0; }
#endif

int pcgc_conv3d_fwd(const float* x, const float* kernel, const float* bias, float* y, int B, int D, int Cin, int Cout,
                    int ksize, int stride, int transposed, int relu, int algo, pcgc_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) return 0;
  PCGC_REQUIRE(x && kernel && y, "pcgc_conv3d_fwd: NULL tensor");
  PCGC_REQUIRE(ksize >= 1 && ksize <= 9 && (ksize & 1) && (stride == 1 || stride == 2) && B >= 0 && D > 0 && Cin > 0 && Cout > 0,
               "pcgc_conv3d_fwd: unsupported geometry k=%d stride=%d", ksize, stride);
  PCGC_REQUIRE(!transposed || stride == 2, "pcgc_conv3d_fwd: transposed conv needs stride=2");
  PCGC_REQUIRE(stride == 1 || transposed || D % 2 == 0, "pcgc_conv3d_fwd: stride-2 conv needs even D");
  if (B == 0) return 0;
  ConvArgs a;
  a.x = x; a.w = kernel; a.bias = bias; a.y = y; a.res = nullptr;
  a.B = B; a.Din = D; a.Dout = transposed ? 2 * D : D / stride;
  a.Cin = Cin; a.Cout = Cout; a.x_cs = Cin; a.x_co = 0; a.y_cs = Cout; a.y_co = 0;
  a.ksize = ksize; a.mode = transposed ? 2 : (stride == 2 ? 1 : 0); a.relu = relu; a.absval = 0; a.lower_bound = 0.f;
  a.w2 = nullptr; a.bias2 = nullptr; a.y2 = nullptr; a.y2_cs = 0; a.cout2 = 0;
  if (algo == 3) {
    int rc = launch_conv_valu(a, s, true);
    PCGC_REQUIRE(rc != 0, "pcgc_conv3d_fwd: no VALU tile kernel for this shape");
    return rc < 0 ? rc : 0;
  }
  if (algo == 0 && (Cin == 1 || Cout == 1)) {            // conv_in / deconv_out: the LDS-tiled VALU kernel, as pcgc_net_forward picks
    const int rc = launch_conv_valu(a, s, true);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (algo != 1 && launch_conv_mfma(a, nullptr, s, false) == 1) {
    float* packed = nullptr;
    const size_t n = mfma_packed_floats(Cin, Cout, ksize, a.mode);
    PCGC_CHECK_HIP(hipMallocAsync((void**)&packed, n * sizeof(float), s));
    int rc = pack_weights_mfma(kernel, packed, Cin, Cout, ksize, a.mode, s);
    if (!rc) { rc = launch_conv_mfma(a, packed, s, true); rc = rc < 0 ? rc : 0; }
    (void)hipFreeAsync(packed, s);
    return rc;
  }
  PCGC_REQUIRE(algo != 2, "pcgc_conv3d_fwd: no MFMA kernel for Cin=%d Cout=%d k=%d stride=%d transposed=%d D=%d", Cin,
               Cout, ksize, stride, transposed, D);
  return launch_conv_direct(a, s);
}

}  // extern "C"
