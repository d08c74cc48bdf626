// UNet executor + C ABI of libetainv_hip.so.
//
// The engine owns: (1) the SD1.x UNet weights, re-laid-out for the kernels (conv [O][tap][I], fused QKV / KV,
// GEGLU row-interleave, one concatenated time-embedding projection), (2) a preallocated NHWC activation
// workspace sized for `max_unet_batch` rows, (3) the attention-map store.  etainv_unet_forward walks the static
// SD1.x graph and only enqueues kernels on the caller's stream: no allocation, no host sync.
// It replaces `self.unet(latent_input, t, encoder_hidden_states=context)["sample"]`
// (reference modules/inversion/eta_inversion.py:321) including what the reference's monkey-patched attention
// forwards do (modules/utils/ptp_utils.py:205-260, modules/utils/masactrl_utils.py:84-127), driven by the
// declarative etainv_attn_ctrl.
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace etainv {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// ---- profiler
struct ProfRec { int cls; double work, bytes; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
static size_t g_prof_used = 0;
static bool g_prof_paused = false;
bool prof_enabled() { return g_prof_on && !g_prof_paused; }
void prof_pause(bool on) { g_prof_paused = on; }
void prof_begin(int cls, double work, hipStream_t s, double bytes) {
  if (g_prof_used == g_prof_pool.size()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    g_prof_pool.emplace_back(a, b);
  }
  auto& ev = g_prof_pool[g_prof_used++];
  g_prof.push_back({cls, work, bytes, ev.first, ev.second});
  (void)hipEventRecord(ev.first, s);
}
void prof_end(hipStream_t s) { (void)hipEventRecord(g_prof.back().b, s); }

enum PackMode { PK_PLAIN = 0, PK_CONV = 1, PK_GEGLU = 2, PK_CONV_IN = 3, PK_CONV_OUT = 4, PK_CONV_IN_GEMM = 5 };

struct WeightSlot {
  std::string name;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0;
  void* dst = nullptr;
  int pack = PK_PLAIN;
  int taps = 1;
  int dst_dtype = ETAINV_F32;  // F32 or the compute dtype (-1 placeholder replaced at build)
  float scale = 1.0f;          // constant folded into the values in fp32 before the rounding to the compute dtype
  void** dst4 = nullptr;       // upsampler convs: also packed as four 2x2 phase kernels (launch_pack_ups4) into *dst4
  float* stage = nullptr;      // fp32 copy kept instead of packing at once: the consumer of a folded LayerNorm is packed when gamma / beta are known
  bool stage_and_pack = false; // fp32 copy kept AND packed (proj_in: the per-image GroupNorm fold needs the fp32 values on every call)
  bool set = false;
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
  }
};

struct Norm { float* g = nullptr; float* b = nullptr; };
struct Lin { void* w = nullptr; float* b = nullptr; int n = 0, k = 0; void* w4 = nullptr; };   // w4: phase form [4][n][4][k] of an upsampler conv (launch_pack_ups4)

struct ResBlock {
  int cin = 0, cout = 0;
  Norm n1, n2;
  Lin conv1, conv2, shortcut;  // conv weights [cout][9][cin]
  int tproj_off = 0;           // column offset into the concatenated time-embedding projection
};
struct TBlock {
  int c = 0;
  Norm gn, ln1, ln2, ln3;
  Lin proj_in, qkv, out1, q, kv, out2, ff1, ff2, proj_out;
  // LayerNorm folded into its consumer (norm1 -> fused QKV, norm2 -> attn2.to_q, norm3 -> GEGLU projection): fp32 staging copies of the
  // consumer weights, the row sums s of the packed gamma-scaled operand and c = beta W^T + bias (kernels.h, launch_ln_fold)
  float *st_qkv = nullptr, *st_q = nullptr, *st_ff1 = nullptr, *st_pin = nullptr;   // st_pin: proj_in, for the per-image GroupNorm fold
  float *s_qkv = nullptr, *c_qkv = nullptr, *s_q = nullptr, *c_q = nullptr, *s_ff1 = nullptr, *c_ff1 = nullptr;
  float q_scale = 1.0f;
};

}  // namespace etainv

using namespace etainv;

struct etainv_engine {
  etainv_engine_config cfg{};
  int dt = ETAINV_F16;
  size_t esz = 2;   // bytes per activation / weight element: 2, or 4 in the fp32-operand mode (compute_dtype ETAINV_F32, f32path.hip)
  int L = 64, maxB = 4, max_img = 1;
  static constexpr int kHeads = 8, kCtx = 77, kCtxDim = 768, kGroups = 32, kTemb = 1280, kCh0 = 320;

  std::vector<WeightSlot> slots;
  std::unordered_map<std::string, int> slot_by_name;
  char* warena = nullptr;
  size_t wbytes = 0;
  char* wsarena = nullptr;
  size_t wsbytes = 0;

  // model
  void* conv_in_w = nullptr; float* conv_in_b = nullptr;
  void* conv_out_w = nullptr; float* conv_out_b = nullptr;
  Norm norm_out;
  Lin time1, time2, tproj;  // tproj: concatenated [sum_co][1280]
  int tproj_total = 0;
  std::vector<ResBlock> res;  // execution order: down (8), mid (2), up (12)
  std::vector<TBlock> tb;     // execution order: 16
  Lin down_conv[3], up_conv[3];

  // workspace (2-byte elements unless noted)
  void *skip[12] = {}, *tmp[3] = {}, *gnbuf = nullptr, *h1 = nullptr, *scbuf = nullptr;
  void *hsA = nullptr, *hsB = nullptr, *lnbuf = nullptr, *qkvbuf = nullptr, *attnbuf = nullptr, *qbuf = nullptr, *kvbuf = nullptr,
       *ffbuf = nullptr, *ctxT = nullptr, *tembuf = nullptr, *temb1 = nullptr, *temb2 = nullptr;
  float *tprojbuf = nullptr, *gn_scratch = nullptr, *maps_acc = nullptr, *lnstat = nullptr, *lnfinal = nullptr;
  size_t maps_bytes = 0;
  // which cross layers the store keeps (etainv_maps_configure): map_div 4 = the five (L/4)^2 layers (default; what LocalBlend and the default eta mask
  // read), 2 = the five (L/2)^2 layers (own lazily allocated buffer, 4x the size), 8 = the mid block's (L/8)^2 layer.  maps_cur = the buffer in use
  int map_div = 4, map_layers = 5;
  float *maps_alt = nullptr, *maps_cur = nullptr;
  size_t maps_alt_bytes = 0, maps_cur_bytes = 0;
  // norm1/2/3 of the transformer blocks folded into the GEMMs around them (no LayerNorm pass over HBM); ETAINV_LN_UNFUSED=1 keeps the
  // standalone LayerNorm kernel (A/B switch, read at engine creation)
  // GroupNorm statistics from the epilogue of the GEMM / conv that wrote the tensor (per-channel partials per wave-tile row block, one buffer per
  // activation buffer a GroupNorm can read: skip[0..11], tmp[0..2], h1) instead of a statistics pass over it; ETAINV_GN_UNFUSED=1 keeps the pass
  // cross-attention K / V of the text context per transformer block.  The context does not change over the 50 steps of a loop: with
  // etainv_engine_cache_context(e, 1) a call whose (context pointer, rows, dtype) equal the previous call's reuses them (16 small GEMMs and the
  // context cast per UNet call, 0.6 % of the benchmark step); the caller switches it off when the loop ends.
  void* kvcache[16] = {};
  bool ctx_cache_on = false;
  const void* ctx_cached = nullptr;
  int ctx_rows = 0, ctx_io = -1;
  void* gn_bufs[16] = {};
  float* gn_part[16] = {};
  float* gn_final = nullptr;
  void* gn_wb = nullptr;
  float* gn_cb = nullptr;
  bool gn_fused = true;
  // ... and, opt-in (ETAINV_GN_FOLD=1), the transformer's GroupNorm (no activation) folded into proj_in through per-image weights: -1.3 % of a
  // 128-row UNet call, nothing measurable on the whole benchmark step -- off by default
  bool gn_fold = false;
  bool ln_fused = true;
  bool ln_folded = false;   // the gamma-scaled consumer weights are packed (redone after any set_weight)
  std::vector<hipStream_t> upload_streams;   // streams weights were uploaded on since the last fold (the fold waits for each that is not the forward's)
  uint64_t ctx_gen = 0, ctx_gen_cached = 0;  // caller's generation of the context buffer (etainv_engine_context_generation)

  // ---- hipGraph replay of small UNet calls (batch 1: ~330 dependent launches of 5 - 30 us).  OPT-IN: the GPU, not the host, paces these calls -- the
  // gap between two dependent kernels (~1.5 us) is the same inside a replayed graph as between eager launches (MI355X_MICROARCH.md, "boundary" row),
  // and the measurement agrees: BASELINE config 2 runs 1.456 images/s with replays and 1.458 without (profiles/r04_cfg2_graph_ab.json).  One captured graph
  // per call signature (rows, latents, io dtype, shared prefix, K / V reuse, attention-control flags); everything that changes from call to call
  // sits behind a FIXED device address: latent / context / output staged through engine-owned buffers, the timesteps in a device vector that a
  // by-value kernel fills in front of the replay.  Calls whose attention control carries per-step device tables (prompt-to-prompt edits) run
  // eagerly, as do calls above ETAINV_GRAPH_MAX_ROWS rows (launch overhead is hidden there) and everything while the event profiler is on.
  void *g_lat = nullptr, *g_ctx = nullptr, *g_out = nullptr;
  float* g_t = nullptr;
  struct GraphEntry { int calls = 0; bool failed = false; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
  std::unordered_map<std::string, GraphEntry> graphs;
  hipStream_t cap_stream = nullptr;   // capture happens on a stream of the engine's own (the caller's is usually the legacy default stream, which cannot capture)
  long long qkv_hm_launches = 0;      // fused QKV projections that wrote the head-major layout since creation
  bool qkv_hm = true;                 // fused QKV projections write head-major planes where the GEMM and the attention kernel both can (ETAINV_QKV_HM)
  int graph_max_rows = 0;             // OFF by default: measured on MI355X at batch 1 (config 2) a replay is exactly as fast as the eager launches (1.456 vs 1.458 images/s)
  int64_t graph_replays = 0, graph_captures = 0;
};

namespace {

struct ArenaPlan {
  size_t off = 0;
  size_t take(size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

// ---- model construction: registers every diffusers parameter name with its destination and packing
struct Builder {
  etainv_engine* e;
  ArenaPlan plan;
  std::vector<std::function<void(char*)>> fixups;  // resolve offsets -> pointers after allocation

  template <typename P>
  void want(P** field, size_t bytes) {
    size_t off = plan.take(bytes);
    fixups.push_back([field, off](char* base) { *field = reinterpret_cast<P*>(base + off); });
  }
  void add_slot(const std::string& name, std::vector<int64_t> shape, void** dst_field_holder, size_t dst_off_bytes, int pack, int taps,
                int dst_dtype) {
    WeightSlot s;
    s.name = name;
    s.ndim = (int)shape.size();
    for (int i = 0; i < s.ndim; ++i) s.shape[i] = shape[i];
    s.pack = pack;
    s.taps = taps;
    s.dst_dtype = dst_dtype;
    int idx = (int)e->slots.size();
    e->slots.push_back(s);
    e->slot_by_name[name] = idx;
    etainv_engine* eng = e;
    fixups.push_back([eng, idx, dst_field_holder, dst_off_bytes](char*) {
      eng->slots[idx].dst = reinterpret_cast<char*>(*dst_field_holder) + dst_off_bytes;
    });
  }
  // the slot keeps an fp32 copy at *holder + off (elements) instead of being packed by set_weight
  void stage_slot(const std::string& name, float** holder, size_t off, bool and_pack = false) {
    const int idx = e->slot_by_name.at(name);
    e->slots[idx].stage_and_pack = and_pack;
    etainv_engine* eng = e;
    fixups.push_back([eng, idx, holder, off](char*) { eng->slots[idx].stage = *holder + off; });
  }
  // fp32 vector parameter (bias / norm scale)
  void vec(const std::string& name, float** field, int n) {
    want(field, (size_t)n * 4);
    add_slot(name, {n}, reinterpret_cast<void**>(field), 0, PK_PLAIN, 1, ETAINV_F32);
  }
  void norm(const std::string& prefix, Norm& nm, int c) {
    vec(prefix + ".weight", &nm.g, c);
    vec(prefix + ".bias", &nm.b, c);
  }
  // linear [n][k] in compute dtype (+ optional fp32 bias)
  void linear(const std::string& prefix, Lin& l, int n, int k, bool bias, int pack = PK_PLAIN) {
    l.n = n;
    l.k = k;
    want(&l.w, (size_t)n * k * e->esz);
    add_slot(prefix + ".weight", {n, k}, &l.w, 0, pack, 1, e->dt);
    if (bias) {
      want(&l.b, (size_t)n * 4);
      add_slot(prefix + ".bias", {n}, reinterpret_cast<void**>(&l.b), 0, pack == PK_GEGLU ? PK_GEGLU : PK_PLAIN, 1, ETAINV_F32);
    }
  }
  void conv1x1(const std::string& prefix, Lin& l, int n, int k) {
    l.n = n;
    l.k = k;
    want(&l.w, (size_t)n * k * e->esz);
    add_slot(prefix + ".weight", {n, k, 1, 1}, &l.w, 0, PK_PLAIN, 1, e->dt);
    want(&l.b, (size_t)n * 4);
    add_slot(prefix + ".bias", {n}, reinterpret_cast<void**>(&l.b), 0, PK_PLAIN, 1, ETAINV_F32);
  }
  void conv3x3(const std::string& prefix, Lin& l, int n, int k) {
    l.n = n;
    l.k = k;
    want(&l.w, (size_t)n * k * 9 * e->esz);
    add_slot(prefix + ".weight", {n, k, 3, 3}, &l.w, 0, PK_CONV, 9, e->dt);
    want(&l.b, (size_t)n * 4);
    add_slot(prefix + ".bias", {n}, reinterpret_cast<void**>(&l.b), 0, PK_PLAIN, 1, ETAINV_F32);
  }
  void conv3x3_ups(const std::string& prefix, Lin& l, int n, int k) {   // conv behind a nearest-2x upsample: the 9-tap form + the four 2x2 phase kernels
    conv3x3(prefix, l, n, k);
    if (e->dt != ETAINV_F32) {
      want(&l.w4, (size_t)16 * n * k * e->esz);
      e->slots[e->slot_by_name.at(prefix + ".weight")].dst4 = &l.w4;
    }
  }
  void resblock(const std::string& prefix, int cin, int cout) {
    e->res.emplace_back();
    // NB: pointers into e->res are taken after all blocks exist (vector may grow) -> reserve() up front
    ResBlock& r = e->res.back();
    r.cin = cin;
    r.cout = cout;
    norm(prefix + ".norm1", r.n1, cin);
    conv3x3(prefix + ".conv1", r.conv1, cout, cin);
    r.tproj_off = e->tproj_total;
    // time_emb_proj rows live inside the concatenated projection matrix
    add_slot(prefix + ".time_emb_proj.weight", {cout, etainv_engine::kTemb}, &e->tproj.w,
             (size_t)r.tproj_off * etainv_engine::kTemb * e->esz, PK_PLAIN, 1, e->dt);
    add_slot(prefix + ".time_emb_proj.bias", {cout}, reinterpret_cast<void**>(&e->tproj.b), (size_t)r.tproj_off * 4, PK_PLAIN, 1,
             ETAINV_F32);
    e->tproj_total += cout;
    norm(prefix + ".norm2", r.n2, cout);
    conv3x3(prefix + ".conv2", r.conv2, cout, cout);
    if (cin != cout) conv1x1(prefix + ".conv_shortcut", r.shortcut, cout, cin);
  }
  void tblock(const std::string& prefix, int c) {
    e->tb.emplace_back();
    TBlock& t = e->tb.back();
    t.c = c;
    const std::string tp = prefix + ".transformer_blocks.0";
    norm(prefix + ".norm", t.gn, c);
    conv1x1(prefix + ".proj_in", t.proj_in, c, c);
    norm(tp + ".norm1", t.ln1, c);
    // fused QKV [3c][c]
    t.qkv.n = 3 * c;
    t.qkv.k = c;
    want(&t.qkv.w, (size_t)3 * c * c * e->esz);
    add_slot(tp + ".attn1.to_q.weight", {c, c}, &t.qkv.w, 0, PK_PLAIN, 1, e->dt);
    // head_dim 40 / 80 (the L^2- and (L/2)^2-token levels): softmax scale * log2(e) folded into to_q, so that the self-attention kernel's
    // score accumulator is directly the exponent argument of exp2 (attention.hip, self_attn40_kernel) -- one rounding of W_q * c instead of
    // W_q, none added
    if (c / etainv_engine::kHeads <= 80 && self_attn40_v2_enabled() && e->dt != ETAINV_F32)
      e->slots.back().scale = (1.0f / std::sqrt((float)(c / etainv_engine::kHeads))) * 1.4426950408889634f;
    t.q_scale = e->slots.back().scale;
    add_slot(tp + ".attn1.to_k.weight", {c, c}, &t.qkv.w, (size_t)c * c * e->esz, PK_PLAIN, 1, e->dt);
    add_slot(tp + ".attn1.to_v.weight", {c, c}, &t.qkv.w, (size_t)2 * c * c * e->esz, PK_PLAIN, 1, e->dt);
    linear(tp + ".attn1.to_out.0", t.out1, c, c, true);
    norm(tp + ".norm2", t.ln2, c);
    linear(tp + ".attn2.to_q", t.q, c, c, false);
    t.kv.n = 2 * c;
    t.kv.k = etainv_engine::kCtxDim;
    want(&t.kv.w, (size_t)2 * c * etainv_engine::kCtxDim * e->esz);
    add_slot(tp + ".attn2.to_k.weight", {c, etainv_engine::kCtxDim}, &t.kv.w, 0, PK_PLAIN, 1, e->dt);
    add_slot(tp + ".attn2.to_v.weight", {c, etainv_engine::kCtxDim}, &t.kv.w, (size_t)c * etainv_engine::kCtxDim * e->esz, PK_PLAIN, 1, e->dt);
    linear(tp + ".attn2.to_out.0", t.out2, c, c, true);
    norm(tp + ".norm3", t.ln3, c);
    linear(tp + ".ff.net.0.proj", t.ff1, 8 * c, c, true, PK_GEGLU);
    linear(tp + ".ff.net.2", t.ff2, c, 4 * c, true);
    conv1x1(prefix + ".proj_out", t.proj_out, c, c);
    if (e->gn_fold && c <= 640) {
      want(&t.st_pin, (size_t)c * c * 4);
      stage_slot(prefix + ".proj_in.weight", &t.st_pin, 0, /*and_pack=*/true);
    }
    if (e->ln_fused) {
      const size_t cc = (size_t)c * c;
      want(&t.st_qkv, 3 * cc * 4);
      want(&t.st_q, cc * 4);
      want(&t.st_ff1, 8 * cc * 4);
      stage_slot(tp + ".attn1.to_q.weight", &t.st_qkv, 0);
      stage_slot(tp + ".attn1.to_k.weight", &t.st_qkv, cc);
      stage_slot(tp + ".attn1.to_v.weight", &t.st_qkv, 2 * cc);
      stage_slot(tp + ".attn2.to_q.weight", &t.st_q, 0);
      stage_slot(tp + ".ff.net.0.proj.weight", &t.st_ff1, 0);
      want(&t.s_qkv, (size_t)3 * c * 4);
      want(&t.c_qkv, (size_t)3 * c * 4);
      want(&t.s_q, (size_t)c * 4);
      want(&t.c_q, (size_t)c * 4);
      want(&t.s_ff1, (size_t)8 * c * 4);
      want(&t.c_ff1, (size_t)8 * c * 4);
    }
  }
};

int build_model(etainv_engine* e) {
  Builder b{e};
  e->res.reserve(22);
  e->tb.reserve(16);
  const int ch[4] = {320, 640, 1280, 1280};
  // conv_in / conv_out keep fp32 weights in their own layouts
  b.want(&e->conv_in_w, (size_t)64 * ch[0] * e->esz);   // [320][64] compute dtype: conv_in runs as a K = 64 GEMM on an im2col buffer
  b.add_slot("conv_in.weight", {ch[0], 4, 3, 3}, &e->conv_in_w, 0, PK_CONV_IN_GEMM, 9, e->dt);
  b.vec("conv_in.bias", &e->conv_in_b, ch[0]);
  b.linear("time_embedding.linear_1", e->time1, etainv_engine::kTemb, ch[0], true);
  b.linear("time_embedding.linear_2", e->time2, etainv_engine::kTemb, etainv_engine::kTemb, true);
  // concatenated time-embedding projection: total = sum of cout over the 22 resblocks = 21120... computed below;
  // reserve the maximum up front (22 * 1280) and trim logically via tproj_total
  b.want(&e->tproj.w, (size_t)22 * 1280 * etainv_engine::kTemb * e->esz);
  b.want(&e->tproj.b, (size_t)22 * 1280 * 4);
  // down
  int cin = ch[0];
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 2; ++j) {
      b.resblock("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? cin : ch[i], ch[i]);
      if (i < 3) b.tblock("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), ch[i]);
    }
    if (i < 3) b.conv3x3("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", e->down_conv[i], ch[i], ch[i]);
    cin = ch[i];
  }
  // mid
  b.resblock("mid_block.resnets.0", 1280, 1280);
  b.tblock("mid_block.attentions.0", 1280);
  b.resblock("mid_block.resnets.1", 1280, 1280);
  // up: rev = (1280,1280,640,320)
  const int rev[4] = {1280, 1280, 640, 320};
  int prev = 1280;
  for (int i = 0; i < 4; ++i) {
    const int cout = rev[i];
    const int cin_skip = rev[std::min(i + 1, 3)];
    for (int j = 0; j < 3; ++j) {
      const int skipc = (j == 2) ? cin_skip : cout;
      const int rin = (j == 0) ? prev : cout;
      b.resblock("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), rin + skipc, cout);
      if (i > 0) b.tblock("up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), cout);
    }
    if (i < 3) b.conv3x3_ups("up_blocks." + std::to_string(i) + ".upsamplers.0.conv", e->up_conv[i], cout, cout);
    prev = cout;
  }
  b.norm("conv_norm_out", e->norm_out, ch[0]);
  b.want(&e->conv_out_w, (size_t)9 * ch[0] * 4 * e->esz);   // [4][9][320] compute dtype: conv_out is an N = 4 implicit GEMM
  b.add_slot("conv_out.weight", {4, ch[0], 3, 3}, &e->conv_out_w, 0, PK_CONV, 9, e->dt);
  b.vec("conv_out.bias", &e->conv_out_b, 4);
  e->tproj.n = e->tproj_total;
  e->tproj.k = etainv_engine::kTemb;

  e->wbytes = b.plan.off;
  ETAINV_HIP(hipMalloc(reinterpret_cast<void**>(&e->warena), e->wbytes));
  ETAINV_HIP(hipMemset(e->warena, 0, e->wbytes));
  // two passes: first resolve `want` fields (pointers), then slot destinations that depend on them
  for (auto& f : b.fixups) f(e->warena);
  for (auto& f : b.fixups) f(e->warena);
  return 0;
}

int build_workspace(etainv_engine* e) {
  ArenaPlan plan;
  std::vector<std::function<void(char*)>> fix;
  auto want = [&](void** field, size_t bytes) {
    size_t off = plan.take(bytes);
    fix.push_back([field, off](char* base) { *field = base + off; });
  };
  const size_t B = (size_t)e->maxB, L = (size_t)e->L, esz = e->esz;
  const size_t hw[4] = {L * L, (L / 2) * (L / 2), (L / 4) * (L / 4), (L / 8) * (L / 8)};
  const size_t skip_el[12] = {hw[0] * 320, hw[0] * 320, hw[0] * 320, hw[1] * 320,  hw[1] * 640,  hw[1] * 640,
                              hw[2] * 640, hw[2] * 1280, hw[2] * 1280, hw[3] * 1280, hw[3] * 1280, hw[3] * 1280};
  for (int i = 0; i < 12; ++i) want(&e->skip[i], B * skip_el[i] * esz);
  // largest activation [HW x C] over the levels is L^2 x 320 (x2 after an upsample conv never exceeds it)
  const size_t hmax = hw[0] * 320;
  // block outputs: the up_blocks.2 upsample conv emits L^2 x 640
  for (int i = 0; i < 3; ++i) want(&e->tmp[i], B * hw[0] * 640 * esz);
  want(&e->gnbuf, B * hw[0] * 960 * esz);
  want(&e->h1, B * hmax * esz);
  want(&e->scbuf, B * hmax * esz);
  want(&e->hsA, B * hmax * esz);
  want(&e->hsB, B * hmax * esz);
  want(&e->lnbuf, B * hmax * esz);
  want(&e->qkvbuf, B * hmax * 3 * esz);
  want(&e->attnbuf, B * hmax * esz);
  want(&e->qbuf, B * hmax * esz);
  want(&e->kvbuf, B * 77 * 2 * 1280 * esz);
  for (size_t i = 0; i < e->tb.size() && i < 16; ++i) want(&e->kvcache[i], B * 77 * 2 * (size_t)e->tb[i].c * esz);
  want(&e->ffbuf, B * hmax * 4 * esz);
  want(&e->ctxT, B * 77 * 768 * esz);
  want(&e->tembuf, B * 320 * esz);
  want(&e->temb1, B * 1280 * esz);
  want(&e->temb2, B * 1280 * esz);
  want(reinterpret_cast<void**>(&e->tprojbuf), B * (size_t)e->tproj_total * 4);
  want(reinterpret_cast<void**>(&e->gn_scratch), B * (GN_MAX_CHUNKS + 1) * etainv_engine::kGroups * 2 * 4);
  // LayerNorm partials [M][P][2]: P = C / (wave tile columns) <= C / 32, M * C <= B * hmax on every level
  want(reinterpret_cast<void**>(&e->lnstat), B * hmax / 32 * 2 * 4);
  want(reinterpret_cast<void**>(&e->lnfinal), B * hw[0] * 2 * 4);   // (mean, rstd) per token row
  // GroupNorm partials: [rows / wm][2][C] floats with wm >= 32 rows per block -> elements / 16 floats per tensor
  for (int i = 0; i < 12; ++i) want(reinterpret_cast<void**>(&e->gn_part[i]), B * skip_el[i] / 16 * 4);
  for (int i = 0; i < 3; ++i) want(reinterpret_cast<void**>(&e->gn_part[12 + i]), B * hw[0] * 640 / 16 * 4);
  want(reinterpret_cast<void**>(&e->gn_part[15]), B * hmax / 16 * 4);
  want(reinterpret_cast<void**>(&e->gn_final), B * (etainv_engine::kGroups * 2 + 2 * 2560) * 4);   // (mean, rstd) + scale / shift planes
  want(&e->gn_wb, B * 640 * 640 * 2);                                   // per-image proj_in weights of a folded GroupNorm (C <= 640)
  want(reinterpret_cast<void**>(&e->gn_cb), B * 640 * 4);
  const size_t res = L / 4;
  e->maps_bytes = (size_t)5 * e->max_img * 2 * etainv_engine::kHeads * res * res * 77 * 4;
  want(reinterpret_cast<void**>(&e->maps_acc), e->maps_bytes);
  want(&e->g_lat, B * 4 * L * L * 4);                                   // graph staging (fp32-sized: the boundary dtype may be fp32)
  want(&e->g_out, B * 4 * L * L * 4);
  want(&e->g_ctx, B * 77 * 768 * 4);
  want(reinterpret_cast<void**>(&e->g_t), B * 4);
  e->wsbytes = plan.off;
  ETAINV_HIP(hipMalloc(reinterpret_cast<void**>(&e->wsarena), e->wsbytes));
  for (auto& f : fix) f(e->wsarena);
  for (int i = 0; i < 12; ++i) e->gn_bufs[i] = e->skip[i];
  for (int i = 0; i < 3; ++i) e->gn_bufs[12 + i] = e->tmp[i];
  e->gn_bufs[15] = e->h1;
  ETAINV_HIP(hipMemset(e->maps_acc, 0, e->maps_bytes));
  e->maps_cur = e->maps_acc;
  e->maps_cur_bytes = e->maps_bytes;
  return 0;
}

// ---- forward helpers
struct Fwd {
  etainv_engine* e;
  hipStream_t s;
  int rows;
  const etainv_attn_ctrl* ctrl;
  int tblock_idx = 0;
  int ctx_rows = 0;        // rows of the context tensor of this call (== n_rows; `rows` shrinks behind a src_exit_block)
  bool kv_reuse = false;   // the K / V projections of this context are already in e->kvcache (etainv_engine_cache_context)

  // rows per GroupNorm partial block of the tensor now in each tracked buffer (0: no partials -- the GroupNorm runs its statistics pass)
  int part_wm[16] = {};
  int part_of(const void* buf) const {
    for (int i = 0; i < 16; ++i)
      if (e->gn_bufs[i] == buf) return i;
    return -1;
  }
  // launch + bookkeeping: gn_out asks the epilogue for the GroupNorm partials of p.out (when the launch shape can: launch_igemm reports the rows
  // per block, 0 otherwise)
  int run(IGemmParams& p, bool gn_out) {
    const int idx = part_of(p.out);
    int wm = 0;
    if (gn_out && e->gn_fused && idx >= 0) {
      p.stat_out = e->gn_part[idx];
      p.stat_kind = 1;
      if (launch_igemm(p, e->dt, s, &wm)) return 1;
    } else if (launch_igemm(p, e->dt, s)) {
      return 1;
    }
    if (idx >= 0) part_wm[idx] = wm;
    return 0;
  }
  // Context-independent prefix (see unet_body): a tensor computed for the first `half` batch rows is copied to rows half .. 2 half - 1, together
  // with the GroupNorm partials its producer left (row blocks are batch-row major: the first half is a prefix of the buffer)
  int dup_rows(void* buf, int have, int n, int hw, int c) {   // rows [0, n) -> rows [have, have + n); a batch row holds hw pixels of c channels
    const size_t elems_per_row = (size_t)hw * c;
    const size_t off = (size_t)have * elems_per_row * e->esz, bytes = (size_t)n * elems_per_row * e->esz;
    ETAINV_HIP(hipMemcpyAsync(reinterpret_cast<char*>(buf) + off, buf, bytes, hipMemcpyDeviceToDevice, s));
    const int idx = part_of(buf);
    if (idx >= 0 && part_wm[idx] > 0) {
      if (hw % part_wm[idx] == 0) {   // [rows * hw / wm][2][C]: a batch row owns hw / wm whole row blocks
        const size_t per_row = (size_t)(hw / part_wm[idx]) * 2 * c * sizeof(float);
        ETAINV_HIP(hipMemcpyAsync(reinterpret_cast<char*>(e->gn_part[idx]) + per_row * have, e->gn_part[idx], per_row * n, hipMemcpyDeviceToDevice, s));
      } else {
        part_wm[idx] = 0;             // row blocks straddle batch rows (small L): no per-row slice to copy -- the consumer runs the statistics pass
      }
    }
    return 0;
  }
  int groupnorm(const void* x1, const void* x2, int c1, int c2, const Norm& nm, int hw, float eps, int silu) {
    const int i1 = part_of(x1), i2 = x2 ? part_of(x2) : -1;
    const int w1 = i1 >= 0 ? part_wm[i1] : 0, w2 = i2 >= 0 ? part_wm[i2] : 0;
    if (w1 > 0 && hw % w1 == 0 && (!x2 || (w2 > 0 && hw % w2 == 0)))
      return launch_groupnorm_pre(x1, x2, c1, c2, e->gn_part[i1], w1, x2 ? e->gn_part[i2] : nullptr, w2, nm.g, nm.b, e->gnbuf, rows, hw,
                                  etainv_engine::kGroups, eps, silu, e->gn_final, e->dt, s);
    return launch_groupnorm(x1, x2, c1, c2, nm.g, nm.b, e->gnbuf, rows, hw, etainv_engine::kGroups, eps, silu, e->gn_scratch, e->dt, s);
  }
  struct LnIn { const float* s; const float* c; };   // folded LayerNorm of the input rows ((mean, rstd) per row in e->lnfinal)
  // ln_out: this GEMM writes the input of a LayerNorm -- leave (mean, rstd) of its output rows in e->lnfinal (partials from the epilogue when
  // the launch can, combined by a small pass; else a pass over the output)
  IGemmParams gemm_params(const void* a, const Lin& l, void* out, int M, const LnIn* ln) {   // plain LayerNorm-consumer GEMM (what gemm() builds for it)
    IGemmParams p;
    p.a1 = a;
    p.w = l.w;
    p.out = out;
    p.M = M;
    p.N = l.n;
    p.c1 = l.k;
    p.H = 1;
    p.W = M;
    p.Ho = 1;
    p.Wo = M;
    p.taps = 1;
    p.rows_per_batch = M;
    p.bias = ln->c;
    p.ln_stat = e->lnfinal;
    p.ln_s = ln->s;
    return p;
  }
  int gemm(const void* a, const Lin& l, void* out, int M, const void* residual = nullptr, int geglu = 0, const void* a2 = nullptr,
           int c1 = 0, int c2 = 0, bool ln_out = false, const LnIn* ln = nullptr, bool gn_out = false, int rows_per_image = 0) {
    IGemmParams p;
    p.a1 = a;
    p.a2 = a2;
    p.w = l.w;
    p.bias = l.b;
    p.residual = residual;
    p.out = out;
    p.M = M;
    p.N = l.n;
    p.c1 = a2 ? c1 : l.k;
    p.c2 = a2 ? c2 : 0;
    p.H = 1;
    p.W = M;
    p.Ho = 1;
    p.Wo = M;
    p.taps = 1;
    p.geglu = geglu;
    p.rows_per_batch = M;
    if (rows_per_image) {   // per-image weights [image][n][k] and bias [image][n]
      p.rows_per_batch = rows_per_image;
      p.w_batch_stride = (int64_t)l.n * l.k;
      p.bias_batch_stride = l.n;
    }
    if (ln) {
      p.bias = ln->c;
      p.ln_stat = e->lnfinal;
      p.ln_s = ln->s;
    }
    if (!ln_out) return run(p, gn_out);
    p.stat_out = e->lnstat;
    int P = 0;
    if (launch_igemm(p, e->dt, s, &P)) return 1;
    if (P == 0) return launch_row_stats(out, e->lnfinal, M, l.n, 1e-5f, e->dt, s);
    return launch_ln_finalize(e->lnstat, P, l.n / P, 1e-5f, e->lnfinal, M, s);
  }
  int conv(const void* a, const Lin& l, void* out, int H, int W, int stride, int ups, const float* rowvec, const void* residual, bool gn_out = true) {
    IGemmParams p;
    p.a1 = a;
    p.w = l.w;
    p.bias = l.b;
    p.rowvec = rowvec;
    p.rowvec_stride = e->tproj_total;
    p.residual = residual;
    p.out = out;
    p.c1 = l.k;
    p.H = H;
    p.W = W;
    p.Ho = ups ? H * 2 : (stride == 2 ? H / 2 : H);
    p.Wo = ups ? W * 2 : (stride == 2 ? W / 2 : W);
    p.stride = stride;
    p.ups = ups;
    p.taps = 9;
    p.M = rows * p.Ho * p.Wo;
    p.N = l.n;
    p.rows_per_batch = p.Ho * p.Wo;
    if (ups && l.w4) {   // four 2 x 2 phase convs on the source image instead of nine taps on the upsampled one (4 / 9 of the FLOPs) where the ring can
      IGemmParams q = p;
      q.ups = 2;
      q.taps = 4;
      q.w = l.w4;
      if (igemm_ups4_ok(q, e->dt)) return run(q, gn_out);
    }
    return run(p, gn_out);
  }
  // x = cat[x1 (c1), x2 (c2)] -> out
  int resblock(const ResBlock& r, const void* x1, const void* x2, int c1, int c2, int side, void* out) {
    const int hw = side * side;
    if (groupnorm(x1, x2, c1, c2, r.n1, hw, 1e-5f, 1)) return 1;
    if (conv(e->gnbuf, r.conv1, e->h1, side, side, 1, 0, e->tprojbuf + r.tproj_off, nullptr)) return 1;
    if (groupnorm(e->h1, nullptr, r.cout, 0, r.n2, hw, 1e-5f, 1)) return 1;
    const void* residual = x1;
    if (r.shortcut.w) {
      if (gemm(x1, r.shortcut, e->scbuf, rows * hw, nullptr, 0, x2, c1, c2)) return 1;
      residual = e->scbuf;
    }
    return conv(e->gnbuf, r.conv2, out, side, side, 1, 0, nullptr, residual);
  }
  // self_rows (0 = all): the sublayers in front of the cross-attention (GroupNorm, proj_in, QKV, self-attention, to_out + residual) do not see
  // the text context; when batch rows r and r + self_rows carry the same latent and timestep (the uncond / cond halves of a CFG call) they are
  // computed for the first self_rows rows only and their result (the residual stream entering the cross-attention) is copied to the other half
  int transformer(const TBlock& t, const void* x, int side, void* out, int self_rows = 0) {
    const int hw = side * side, c = t.c, d = c / etainv_engine::kHeads;
    const int blk = tblock_idx++;
    const bool fold = e->ln_fused;
    int mode = 0, n_img = 1;
    if (ctrl) {
      n_img = ctrl->n_img;
      if (ctrl->mode == ETAINV_ATTN_PTP && ctrl->self_replace_active && hw <= ctrl->self_max_tokens) mode = 1;
      if (ctrl->mode == ETAINV_ATTN_MASA && ctrl->masa_active && blk >= ctrl->masa_first_block) mode = 2;
    }
    const int all_rows = rows;
    if (self_rows <= 0 || self_rows >= all_rows || 2 * self_rows < all_rows || mode != 0 || e->gn_fold) self_rows = all_rows;   // (a row remap couples the halves: no sharing)
    rows = self_rows;
    int M = rows * hw;
    // The GroupNorm in front of proj_in has no activation: with its statistics known from the producer's epilogue it becomes a per-image scaling
    // of proj_in's input channels + a per-image bias -- folded into per-image copies of the (small) weight matrix instead of a pass over x.
    // Levels with C <= 640 (C = 1280: the 128 weight copies would cost more than the pass) and images that are whole M tiles.
    const int ix = part_of(x), wx = ix >= 0 ? part_wm[ix] : 0;
    if (e->gn_fold && t.st_pin && wx > 0 && hw % wx == 0 && hw % 256 == 0) {
      if (launch_gn_finalize(c, 0, e->gn_part[ix], wx, nullptr, 0, rows, hw, etainv_engine::kGroups, 1e-6f, e->gn_final, s)) return 1;
      if (launch_gn_fold(t.st_pin, t.gn.g, t.gn.b, t.proj_in.b, e->gn_final, etainv_engine::kGroups, rows, c, c, e->gn_wb, e->gn_cb, e->dt, s)) return 1;
      Lin pin;
      pin.w = e->gn_wb;
      pin.b = e->gn_cb;
      pin.n = c;
      pin.k = c;
      if (gemm(x, pin, e->hsA, M, nullptr, 0, nullptr, 0, 0, fold, nullptr, false, hw)) return 1;
    } else {
      if (groupnorm(x, nullptr, c, 0, t.gn, hw, 1e-6f, 0)) return 1;
      if (gemm(e->gnbuf, t.proj_in, e->hsA, M, nullptr, 0, nullptr, 0, 0, fold)) return 1;
    }
    // self-attention
    int head_major = 0;   // the fused QKV projection writes head-major planes when both sides can (section 4.2: a 64-key tile becomes one contiguous block)
    if (fold) {
      const LnIn ln1{t.s_qkv, t.c_qkv};
      if (e->qkv_hm && self_attn_head_major_ok(d, e->dt)) {
        IGemmParams q = gemm_params(e->hsA, t.qkv, e->qkvbuf, M, &ln1);
        q.hm_heads = etainv_engine::kHeads;
        q.hm_dim = d;
        q.hm_tokens = hw;
        if (igemm_hm_ok(q, e->dt)) {
          head_major = 1;
          ++e->qkv_hm_launches;
          if (run(q, false)) return 1;
        }
      }
      if (!head_major && gemm(e->hsA, t.qkv, e->qkvbuf, M, nullptr, 0, nullptr, 0, 0, false, &ln1)) return 1;
    } else {
      if (launch_layernorm(e->hsA, t.ln1.g, t.ln1.b, e->lnbuf, M, c, 1e-5f, e->dt, s)) return 1;
      if (gemm(e->lnbuf, t.qkv, e->qkvbuf, M)) return 1;
    }
    if (launch_self_attention_mode(e->qkvbuf, e->attnbuf, rows, hw, etainv_engine::kHeads, d, mode, n_img, e->dt, s, /*q_prescaled=*/d <= 80 && self_attn40_v2_enabled() && e->dt != ETAINV_F32,
                                   ctrl ? (ctrl->src_exit_block ? -1 : ctrl->first_row) : 0, head_major)) return 1;
    if (gemm(e->attnbuf, t.out1, e->hsB, M, e->hsA, 0, nullptr, 0, 0, fold)) return 1;
    if (self_rows != all_rows) {   // the other half of the batch enters the cross-attention with the same residual stream (and LayerNorm statistics)
      const size_t off = (size_t)M * c * e->esz, Mn = (size_t)(all_rows - self_rows) * hw;   // rows [0, all - self) -> rows [self, all)
      ETAINV_HIP(hipMemcpyAsync(reinterpret_cast<char*>(e->hsB) + off, e->hsB, Mn * c * e->esz, hipMemcpyDeviceToDevice, s));
      if (fold) ETAINV_HIP(hipMemcpyAsync(e->lnfinal + (size_t)M * 2, e->lnfinal, Mn * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
      rows = all_rows;
      M = rows * hw;
    }
    // cross-attention
    if (fold) {
      const LnIn ln2{t.s_q, t.c_q};
      if (gemm(e->hsB, t.q, e->qbuf, M, nullptr, 0, nullptr, 0, 0, false, &ln2)) return 1;
    } else {
      if (launch_layernorm(e->hsB, t.ln2.g, t.ln2.b, e->lnbuf, M, c, 1e-5f, e->dt, s)) return 1;
      if (gemm(e->lnbuf, t.q, e->qbuf, M)) return 1;
    }
    void* kvb = e->kvcache[blk];
    // (all rows of the call's context, also behind a src_exit_block: a later call that reuses the cache may exit later)
    if (!kv_reuse && gemm(e->ctxT, t.kv, kvb, ctx_rows * etainv_engine::kCtx)) return 1;
    CrossParams cp;
    cp.N = hw;
    cp.heads = etainv_engine::kHeads;
    cp.n_ctx = etainv_engine::kCtx;
    cp.scale_log2 = (1.0f / std::sqrt((float)d)) * 1.4426950408889634f;
    cp.rows = rows;
    cp.n_img_cap = e->max_img;
    cp.maps_acc = e->maps_cur;
    if (ctrl && (ctrl->mode == ETAINV_ATTN_PTP || ctrl->mode == ETAINV_ATTN_STORE)) {
      cp.n_img = ctrl->n_img;
      cp.layout = ctrl->mode == ETAINV_ATTN_PTP ? (ctrl->src_exit_block ? 3 : 2) : 1;   // 3: rows [u_t, c_t, c_s] (cond source rows leave early)
      cp.first_row = ctrl->mode == ETAINV_ATTN_PTP ? ctrl->first_row : 0;
      if (ctrl->mode == ETAINV_ATTN_PTP && ctrl->cross_alpha && (ctrl->mapper || ctrl->replace_mat)) {
        cp.edit = 1;
        cp.mapper = ctrl->mapper;
        cp.alphas = ctrl->alphas;
        cp.replace_mat = ctrl->replace_mat;
        cp.equalizer = ctrl->equalizer;
        cp.cross_alpha = ctrl->cross_alpha;
      }
      const int res = e->L / e->map_div;
      if (ctrl->store_maps && hw == res * res) {
        // the five (L/4)^2-token cross layers: transformer blocks 4,5 (down) and 7,8,9 (up); (L/2)^2: 2,3 and 10,11,12; (L/8)^2: the mid block 6
        static const int layer_of_block[3][16] = {{-1, -1, 0, 1, -1, -1, -1, -1, -1, -1, 2, 3, 4, -1, -1, -1},
                                                  {-1, -1, -1, -1, 0, 1, -1, 2, 3, 4, -1, -1, -1, -1, -1, -1},
                                                  {-1, -1, -1, -1, -1, -1, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1}};
        cp.map_layer = layer_of_block[e->map_div == 2 ? 0 : e->map_div == 4 ? 1 : 2][blk];
      }
    }
    if (launch_cross_attention_p(e->qbuf, kvb, e->attnbuf, rows, d, cp, e->dt, s)) return 1;
    if (gemm(e->attnbuf, t.out2, e->hsA, M, e->hsB, 0, nullptr, 0, 0, fold)) return 1;
    // feed-forward (GEGLU)
    if (fold) {
      const LnIn ln3{t.s_ff1, t.c_ff1};
      if (gemm(e->hsA, t.ff1, e->ffbuf, M, nullptr, 1, nullptr, 0, 0, false, &ln3)) return 1;
    } else {
      if (launch_layernorm(e->hsA, t.ln3.g, t.ln3.b, e->lnbuf, M, c, 1e-5f, e->dt, s)) return 1;
      if (gemm(e->lnbuf, t.ff1, e->ffbuf, M, nullptr, 1)) return 1;
    }
    if (gemm(e->ffbuf, t.ff2, e->hsB, M, e->hsA)) return 1;
    return gemm(e->hsB, t.proj_out, out, M, x, 0, nullptr, 0, 0, false, nullptr, /*gn_out=*/true);
  }
};

void* pick_tmp(etainv_engine* e, const void* a, const void* b) {
  for (int i = 0; i < 3; ++i)
    if (e->tmp[i] != a && e->tmp[i] != b) return e->tmp[i];
  return nullptr;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" int etainv_abi_version(void) { return ETAINV_ABI_VERSION; }
extern "C" const char* etainv_last_error(void) { return g_err.c_str(); }

extern "C" int etainv_engine_create(const etainv_engine_config* cfg, etainv_engine_t** out) {
  ETAINV_CHECK(cfg && out, "null argument");
  ETAINV_CHECK(cfg->compute_dtype == ETAINV_F16 || cfg->compute_dtype == ETAINV_BF16 || cfg->compute_dtype == ETAINV_F32, "compute_dtype must be f16, bf16 or f32");
  ETAINV_CHECK(cfg->latent_size >= 8 && cfg->latent_size % 8 == 0 && cfg->latent_size <= 128, "latent_size must be a multiple of 8 in [8,128]");
  ETAINV_CHECK(cfg->max_unet_batch >= 1 && cfg->max_img >= 1, "batch sizes must be positive");
  int ndev = 0;
  ETAINV_HIP(hipGetDeviceCount(&ndev));
  ETAINV_CHECK(ndev > 0, "no HIP device visible: the etainv engine has no CPU fallback");
  auto* e = new etainv_engine();
  e->cfg = *cfg;
  e->dt = cfg->compute_dtype;
  e->L = cfg->latent_size;
  e->maxB = cfg->max_unet_batch;
  e->max_img = cfg->max_img;
  e->esz = e->dt == ETAINV_F32 ? 4 : 2;
  // fp32-operand mode: the standalone norms (the folds are fusions of the 16-bit kernels' epilogues)
  e->ln_fused = !env_on("ETAINV_LN_UNFUSED") && e->dt != ETAINV_F32;
  e->gn_fused = !env_on("ETAINV_GN_UNFUSED") && e->dt != ETAINV_F32;
  e->gn_fold = e->gn_fused && env_on("ETAINV_GN_FOLD");
  // hipGraph replay of small calls: opt-in, ETAINV_GRAPH_MAX_ROWS=<rows> (calls of at most that many UNet rows are captured and replayed)
  if (const char* gm = getenv("ETAINV_GRAPH_MAX_ROWS")) e->graph_max_rows = atoi(gm);
  e->qkv_hm = env_flag("ETAINV_QKV_HM", true);   // (default on: +0.8 % on the benchmark step, +1 % on config 5; "0" = row-major)
  if (build_model(e) || build_workspace(e)) {
    etainv_engine_destroy(e);
    return 1;
  }
  *out = e;
  return 0;
}

extern "C" int etainv_engine_destroy(etainv_engine_t* e) {
  if (!e) return 0;
  for (auto& kv : e->graphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
  }
  if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
  if (e->warena) (void)hipFree(e->warena);
  if (e->maps_alt) (void)hipFree(e->maps_alt);
  if (e->wsarena) (void)hipFree(e->wsarena);
  delete e;
  return 0;
}

extern "C" int etainv_engine_num_weights(etainv_engine_t* e) { return e ? (int)e->slots.size() : 0; }

extern "C" int etainv_engine_weight_info(etainv_engine_t* e, int i, char* name, int name_cap, int64_t shape[4], int* ndim) {
  ETAINV_CHECK(e && i >= 0 && i < (int)e->slots.size() && name && shape && ndim, "bad arguments");
  const WeightSlot& s = e->slots[i];
  ETAINV_CHECK((int)s.name.size() + 1 <= name_cap, "name buffer too small");
  std::memcpy(name, s.name.c_str(), s.name.size() + 1);
  for (int k = 0; k < 4; ++k) shape[k] = s.shape[k];
  *ndim = s.ndim;
  return 0;
}

extern "C" int etainv_engine_set_weight(etainv_engine_t* e, const char* name, const float* data, int64_t numel, void* stream) {
  ETAINV_CHECK(e && name && data, "null argument");
  auto it = e->slot_by_name.find(name);
  ETAINV_CHECK(it != e->slot_by_name.end(), std::string("unknown parameter ") + name);
  WeightSlot& s = e->slots[it->second];
  ETAINV_CHECK(numel == s.numel(), std::string("size mismatch for ") + name);
  int64_t rows = s.shape[0], cols = s.numel() / s.shape[0];
  if (s.ndim == 1) { rows = s.shape[0]; cols = 1; }
  if (s.stage) ETAINV_HIP(hipMemcpyAsync(s.stage, data, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if ((!s.stage || s.stage_and_pack) && launch_pack_weight(data, s.dst, rows, cols, s.pack, s.taps, s.dst_dtype, (hipStream_t)stream, s.scale)) return 1;
  if (s.dst4 && launch_pack_ups4(data, *s.dst4, (int)s.shape[0], (int)s.shape[1], s.dst_dtype, (hipStream_t)stream)) return 1;
  s.set = true;
  e->ln_folded = false;
  bool seen = false;
  for (hipStream_t u : e->upload_streams) seen = seen || u == (hipStream_t)stream;
  if (!seen) e->upload_streams.push_back((hipStream_t)stream);
  return 0;
}

// gamma / beta of norm1/2/3 folded into the staged consumer weights of every transformer block (launch_ln_fold); runs on the forward's stream
// before the first UNet call after the weights changed
static int fold_layernorms(etainv_engine* e, hipStream_t s) {
  for (hipStream_t u : e->upload_streams)
    if (u != s) ETAINV_HIP(hipStreamSynchronize(u));
  e->upload_streams.clear();
  for (const TBlock& t : e->tb) {
    const int64_t c = t.c, cc = c * c;
    for (int part = 0; part < 3; ++part)
      if (launch_ln_fold(t.st_qkv + part * cc, t.ln1.g, t.ln1.b, nullptr, c, c, PK_PLAIN, part == 0 ? t.q_scale : 1.0f,
                         reinterpret_cast<char*>(t.qkv.w) + (size_t)part * cc * 2, t.s_qkv + part * c, t.c_qkv + part * c, e->dt, s))
        return 1;
    if (launch_ln_fold(t.st_q, t.ln2.g, t.ln2.b, nullptr, c, c, PK_PLAIN, 1.0f, t.q.w, t.s_q, t.c_q, e->dt, s)) return 1;
    if (launch_ln_fold(t.st_ff1, t.ln3.g, t.ln3.b, t.ff1.b, 8 * c, c, PK_GEGLU, 1.0f, t.ff1.w, t.s_ff1, t.c_ff1, e->dt, s)) return 1;
  }
  e->ln_folded = true;
  return 0;
}

extern "C" int etainv_engine_weights_ready(etainv_engine_t* e) {
  if (!e) return 0;
  for (auto& s : e->slots)
    if (!s.set) return 0;
  return 1;
}

extern "C" int etainv_engine_cache_context(etainv_engine_t* e, int enable) {
  ETAINV_CHECK(e, "null engine");
  e->ctx_cache_on = enable != 0;
  e->ctx_cached = nullptr;   // (the first call after switching it on computes the projections)
  return 0;
}

extern "C" int etainv_engine_context_generation(etainv_engine_t* e, uint64_t generation) {
  ETAINV_CHECK(e, "null engine");
  e->ctx_gen = generation;
  return 0;
}

extern "C" int64_t etainv_engine_workspace_bytes(etainv_engine_t* e) { return e ? (int64_t)e->wsbytes : 0; }
extern "C" int64_t etainv_engine_weight_bytes(etainv_engine_t* e) { return e ? (int64_t)e->wbytes : 0; }

static int unet_body(etainv_engine_t* e, const void* latent, int n_lat, const int64_t* t_host, const void* ctx, int n_rows,
                     const etainv_attn_ctrl* ctrl, void* out, int io_dtype, void* stream, const float* t_dev, int kv_reuse_forced);

static size_t io_size(int dt) { return dt == ETAINV_F32 ? 4 : 2; }

// Replay of a captured call (see etainv_engine::graphs).  Returns 0 = replayed, 1 = error (message set), 2 = not taken: run the call eagerly.
static int unet_graph(etainv_engine_t* e, const void* latent, int n_lat, const int64_t* t_host, const void* ctx, int n_rows, const etainv_attn_ctrl* ctrl,
                      void* out, int io_dtype, void* stream) {
  if (e->graph_max_rows <= 0 || n_rows > e->graph_max_rows || prof_enabled()) return 2;
  if (ctrl && (ctrl->mapper || ctrl->alphas || ctrl->replace_mat || ctrl->equalizer || ctrl->cross_alpha)) return 2;   // per-step device tables
  hipStream_t s = (hipStream_t)stream;
  if (e->ln_fused && !e->ln_folded && fold_layernorms(e, s)) return 1;   // (host work with stream synchronisation: never inside a capture)
  bool t_pairs = n_rows > n_lat && n_rows <= 2 * n_lat;                   // what unet_body's shared-prefix decision reads from the timesteps
  for (int r = n_lat; t_pairs && r < n_rows; ++r) t_pairs = t_host[r] == t_host[r - n_lat];
  const bool reuse = e->ctx_cache_on && e->ctx_cached == ctx && e->ctx_rows == n_rows && e->ctx_io == io_dtype && e->ctx_gen_cached == e->ctx_gen;
  std::string key;   // (built with std::to_string: no fixed buffer a long signature could truncate into another signature's key)
  for (const int v : {n_rows, n_lat, io_dtype, (int)t_pairs, (int)reuse, e->map_div, ctrl ? ctrl->mode : -1, ctrl ? ctrl->n_img : 0, ctrl ? ctrl->store_maps : 0,
                      ctrl ? ctrl->self_replace_active : 0, ctrl ? ctrl->self_max_tokens : 0, ctrl ? ctrl->masa_active : 0, ctrl ? ctrl->masa_first_block : 0,
                      ctrl ? ctrl->first_row : 0, ctrl ? ctrl->src_exit_block : 0}) {
    key += std::to_string(v);
    key += '.';
  }
  auto& ge = e->graphs[key];
  // the first call of a signature runs eagerly: one-time allocations, function attributes and the LayerNorm fold happen outside any capture
  if (ge.failed || ++ge.calls < 2) return 2;
  const size_t lat_bytes = (size_t)n_lat * 4 * e->L * e->L * io_size(io_dtype), out_bytes = (size_t)n_rows * 4 * e->L * e->L * io_size(io_dtype);
  ETAINV_HIP(hipMemcpyAsync(e->g_lat, latent, lat_bytes, hipMemcpyDeviceToDevice, s));
  if (!reuse) ETAINV_HIP(hipMemcpyAsync(e->g_ctx, ctx, (size_t)n_rows * etainv_engine::kCtx * etainv_engine::kCtxDim * io_size(io_dtype), hipMemcpyDeviceToDevice, s));
  if (launch_set_timesteps(t_host, n_rows, e->g_t, s)) return 1;
  if (!ge.exec) {
    // (a stream that cannot be created or put into capture mode sends this signature to the eager path, like every other capture failure)
    if ((!e->cap_stream && hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) ||
        hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
      (void)hipGetLastError();
      ge.failed = true;
      return 2;
    }
    const int rc = unet_body(e, e->g_lat, n_lat, t_host, e->g_ctx, n_rows, ctrl, e->g_out, io_dtype, (void*)e->cap_stream, e->g_t, reuse ? 1 : 0);
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(e->cap_stream, &g);
    if (rc || ec != hipSuccess || !g || hipGraphInstantiate(&ge.exec, g, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (g) (void)hipGraphDestroy(g);
      ge.exec = nullptr;
      ge.failed = true;                    // this signature stays on the eager path (the staged inputs are untouched copies)
      return 2;
    }
    ge.graph = g;
    ++e->graph_captures;
  }
  ETAINV_HIP(hipGraphLaunch(ge.exec, s));
  ETAINV_HIP(hipMemcpyAsync(out, e->g_out, out_bytes, hipMemcpyDeviceToDevice, s));
  ++e->graph_replays;
  return 0;
}

extern "C" int etainv_unet_forward(etainv_engine_t* e, const void* latent, int n_lat, const int64_t* t_host, const void* ctx, int n_rows,
                                   const etainv_attn_ctrl* ctrl, void* out, int io_dtype, void* stream) {
  int rc = 2;
  if (e && latent && t_host && ctx && out && n_rows >= 1 && n_rows <= e->maxB && n_lat >= 1 && n_lat <= n_rows && etainv_engine_weights_ready(e))
    rc = unet_graph(e, latent, n_lat, t_host, ctx, n_rows, ctrl, out, io_dtype, stream);
  if (rc == 1) return 1;
  if (rc == 2 && unet_body(e, latent, n_lat, t_host, ctx, n_rows, ctrl, out, io_dtype, stream, nullptr, -1)) return 1;
  if (e->ctx_cache_on) {
    e->ctx_cached = ctx;
    e->ctx_rows = n_rows;
    e->ctx_io = io_dtype;
    e->ctx_gen_cached = e->ctx_gen;
  }
  return 0;
}

/* fused QKV projections that wrote the head-major planes since the engine was created (tests: the layout is really on the path) */
extern "C" int etainv_engine_qkv_head_major_count(etainv_engine_t* e, long long* launches) {
  ETAINV_CHECK(e && launches, "bad arguments");
  *launches = e->qkv_hm_launches;
  return 0;
}

extern "C" int etainv_engine_graph_stats(etainv_engine_t* e, int64_t* captures, int64_t* replays) {
  ETAINV_CHECK(e && captures && replays, "null argument");
  *captures = e->graph_captures;
  *replays = e->graph_replays;
  return 0;
}

static int unet_body(etainv_engine_t* e, const void* latent, int n_lat, const int64_t* t_host, const void* ctx, int n_rows,
                     const etainv_attn_ctrl* ctrl, void* out, int io_dtype, void* stream, const float* t_dev, int kv_reuse_forced) {
  ETAINV_CHECK(e && latent && t_host && ctx && out, "null argument");
  ETAINV_CHECK(n_rows >= 1 && n_rows <= e->maxB, "n_rows exceeds max_unet_batch");
  ETAINV_CHECK(n_lat >= 1 && n_lat <= n_rows, "1 <= n_lat <= n_rows (UNet row r reads latent r % n_lat)");
  ETAINV_CHECK(etainv_engine_weights_ready(e), "weights not fully set");
  if (ctrl) {
    ETAINV_CHECK(ctrl->n_img >= 1 && ctrl->n_img <= e->max_img, "ctrl.n_img exceeds max_img");
    ETAINV_CHECK(ctrl->first_row == 0 || (ctrl->mode == ETAINV_ATTN_PTP && ctrl->first_row == ctrl->n_img), "first_row: 0, or n_img with prompt-to-prompt");
    // (the self-replace reaches up to the (L/2)^2-token layers: transformer blocks <= 12 -- an exit in front of that would starve it)
    ETAINV_CHECK(ctrl->src_exit_block == 0 || (ctrl->mode == ETAINV_ATTN_PTP && ctrl->first_row == ctrl->n_img && !ctrl->mapper && !ctrl->replace_mat &&
                                               ctrl->src_exit_block >= (ctrl->self_replace_active ? 12 : 9) && ctrl->src_exit_block < 15),
                 "src_exit_block: prompt-to-prompt three-row call without cross edit; exit after block 9..14 (12..14 while the self-replace is active)");
    // the store may keep other layers than the (L/4)^2 ones (etainv_maps_configure): an exit in front of the last stored block would silently drop the
    // source rows' maps of the later layers (last stored block: res_div 2 -> 12, 4 -> 9, 8 -> 6)
    ETAINV_CHECK(!ctrl->src_exit_block || !ctrl->store_maps || (e->map_div == 2 ? 12 : e->map_div == 4 ? 9 : 6) <= ctrl->src_exit_block,
                 "src_exit_block lies in front of the last cross-attention layer the map store keeps (etainv_maps_configure): the source rows' maps would be lost");
    if (ctrl->mode == ETAINV_ATTN_PTP || ctrl->mode == ETAINV_ATTN_MASA)
      ETAINV_CHECK(n_rows == 4 * ctrl->n_img - ctrl->first_row, "ptp / masactrl need 4*n_img UNet rows [u_s,u_t,c_s,c_t] (ptp with first_row = n_img: 3*n_img rows [u_t,c_s,c_t])");
    if (ctrl->mode == ETAINV_ATTN_STORE)
      ETAINV_CHECK(n_rows == 2 * ctrl->n_img || n_rows == ctrl->n_img, "store mode needs n_img or 2*n_img rows");
  }
  hipStream_t s = (hipStream_t)stream;
  const int L = e->L;
  if (e->ln_fused && !e->ln_folded && fold_layernorms(e, s)) return 1;
  Fwd f{e, s, n_rows, ctrl};
  f.ctx_rows = n_rows;

  // timesteps (by value in the kernel arguments: no host buffer outlives this call) -> embedding, MLP, all 22 projections in one GEMM
  // (t_dev: the graph path -- the timesteps of a replay are whatever launch_set_timesteps wrote to the device vector in front of it)
  if (t_dev ? launch_time_embedding_dev(t_dev, n_rows, etainv_engine::kCh0, e->tembuf, e->dt, s)
            : launch_time_embedding(t_host, n_rows, etainv_engine::kCh0, e->tembuf, e->dt, s)) return 1;
  if (f.gemm(e->tembuf, e->time1, e->temb1, n_rows)) return 1;
  if (launch_silu(e->temb1, e->temb1, (int64_t)n_rows * etainv_engine::kTemb, e->dt, s)) return 1;
  if (f.gemm(e->temb1, e->time2, e->temb2, n_rows)) return 1;
  if (launch_silu(e->temb2, e->temb2, (int64_t)n_rows * etainv_engine::kTemb, e->dt, s)) return 1;
  {
    IGemmParams p;
    p.a1 = e->temb2;
    p.w = e->tproj.w;
    p.bias = e->tproj.b;
    p.out = e->tprojbuf;
    p.out_f32 = 1;
    p.M = n_rows;
    p.N = e->tproj_total;
    p.c1 = etainv_engine::kTemb;
    p.W = n_rows;
    p.Wo = n_rows;
    p.rows_per_batch = n_rows;
    if (launch_igemm(p, e->dt, s)) return 1;
  }
  f.kv_reuse = kv_reuse_forced >= 0 ? kv_reuse_forced != 0
                                    : e->ctx_cache_on && e->ctx_cached == ctx && e->ctx_rows == n_rows && e->ctx_io == io_dtype && e->ctx_gen_cached == e->ctx_gen;
  // the cache entry becomes valid only when this forward has enqueued every projection (set at the end of unet_body): an error half-way
  // leaves it invalid, so the next call projects again
  e->ctx_cached = nullptr;
  if (!f.kv_reuse && launch_cast_f32(ctx, io_dtype, e->ctxT, e->dt, (int64_t)n_rows * etainv_engine::kCtx * etainv_engine::kCtxDim, s)) return 1;

  // ---- down path
  // Context-independent prefix: with classifier-free guidance the call carries every latent twice (rows r and r + n_lat: uncond / cond context,
  // same latent, same timestep).  Nothing in front of the first cross-attention reads the context -- conv_in, the first residual block and the first
  // transformer block's GroupNorm / proj_in / QKV / self-attention (N = L^2: the most expensive launch of the call) / to_out -- so those run on
  // n_lat rows and their results are copied to the other half (3 device copies, ~0.3 ms, for 2.6 % of a 128-row call).  The reference evaluates
  // both halves (eta_inversion.py:320-321 `torch.cat([latent] * 2)`); the values are the same.  ETAINV_NO_PREFIX_SHARE=1: A/B switch.
  // (general form: n_lat < n_rows <= 2 n_lat -- rows r >= n_lat repeat latent r - n_lat; the 3 n_img-row backward calls of eta == 0 steps carry
  // latents [tgt, src] and rows [u_t, c_s, c_t]: the last n_img rows repeat the first n_img)
  bool share = !env_on("ETAINV_NO_PREFIX_SHARE") && n_rows > n_lat && n_rows <= 2 * n_lat && !e->gn_fold;
  for (int r = n_lat; share && r < n_rows; ++r) share = t_host[r] == t_host[r - n_lat];
  const int dup_n = n_rows - n_lat;   // rows copied from the head of every shared tensor to its tail
  const int pre_rows = share ? n_lat : n_rows;
  f.rows = pre_rows;
  if (launch_im2col_in(latent, io_dtype, n_lat, pre_rows, L, e->gnbuf, e->dt, s)) return 1;
  {
    Lin cin_l;
    cin_l.w = e->conv_in_w;
    cin_l.b = e->conv_in_b;
    cin_l.n = etainv_engine::kCh0;
    cin_l.k = 64;
    if (f.gemm(e->gnbuf, cin_l, e->skip[0], pre_rows * L * L, nullptr, 0, nullptr, 0, 0, false, nullptr, /*gn_out=*/true)) return 1;
  }
  const int ch[4] = {320, 640, 1280, 1280};
  int ri = 0, ti = 0, si = 1, side = L;
  const void* h = e->skip[0];
  int hc = 320;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 2; ++j) {
      if (i < 3) {
        void* r_out = pick_tmp(e, h, nullptr);
        if (f.resblock(e->res[ri++], h, nullptr, hc, 0, side, r_out)) return 1;
        const bool first = i == 0 && j == 0;
        if (first && share) {   // skip[0] (a skip connection of the up path) and the block input (proj_out's residual) are needed for all rows
          if (f.dup_rows(e->skip[0], pre_rows, dup_n, L * L, etainv_engine::kCh0) || f.dup_rows(r_out, pre_rows, dup_n, L * L, ch[0])) return 1;
        }
        f.rows = n_rows;
        if (f.transformer(e->tb[ti++], r_out, side, e->skip[si], first && share ? pre_rows : 0)) return 1;
      } else {
        if (f.resblock(e->res[ri++], h, nullptr, hc, 0, side, e->skip[si])) return 1;
      }
      h = e->skip[si++];
      hc = ch[i];
    }
    if (i < 3) {
      if (f.conv(h, e->down_conv[i], e->skip[si], side, side, 2, 0, nullptr, nullptr)) return 1;
      h = e->skip[si++];
      side /= 2;
    }
  }
  // ---- mid
  {
    void* a = pick_tmp(e, h, nullptr);
    if (f.resblock(e->res[ri++], h, nullptr, 1280, 0, side, a)) return 1;
    void* b2 = pick_tmp(e, a, nullptr);
    if (f.transformer(e->tb[ti++], a, side, b2)) return 1;
    void* c = pick_tmp(e, b2, nullptr);
    if (f.resblock(e->res[ri++], b2, nullptr, 1280, 0, side, c)) return 1;
    h = c;
    hc = 1280;
  }
  // ---- up path
  const int rev[4] = {1280, 1280, 640, 320};
  const int skip_c[12] = {320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280};
  int sp = 11;
  for (int i = 0; i < 4; ++i) {
    const int cout = rev[i];
    for (int j = 0; j < 3; ++j) {
      const void* sk = e->skip[sp];
      const int skc = skip_c[sp];
      --sp;
      void* r_out = pick_tmp(e, h, nullptr);
      if (f.resblock(e->res[ri++], h, sk, hc, skc, side, r_out)) return 1;
      h = r_out;
      hc = cout;
      if (i > 0) {
        void* t_out = pick_tmp(e, h, nullptr);
        if (f.transformer(e->tb[ti++], h, side, t_out)) return 1;
        h = t_out;
        // the cond source rows (last n_img of [u_t, c_t, c_s]) have fed their last stored attention layer: everything after runs on the other rows
        if (ctrl && ctrl->src_exit_block && ti == ctrl->src_exit_block + 1) f.rows = 2 * ctrl->n_img;
      }
    }
    if (i < 3) {
      void* u = pick_tmp(e, h, nullptr);
      if (f.conv(h, e->up_conv[i], u, side, side, 1, 1, nullptr, nullptr)) return 1;
      h = u;
      side *= 2;
    }
  }
  // ---- out
  if (f.groupnorm(h, nullptr, 320, 0, e->norm_out, L * L, 1e-5f, 1)) return 1;
  {
    IGemmParams p;
    p.a1 = e->gnbuf;
    p.w = e->conv_out_w;
    p.bias = e->conv_out_b;
    p.out = out;
    p.out_nchw = 4;
    p.out_io_dtype = io_dtype;
    p.c1 = 320;
    p.H = p.W = p.Ho = p.Wo = L;
    p.taps = 9;
    p.M = f.rows * L * L;
    p.N = 4;
    p.rows_per_batch = L * L;
    return launch_igemm(p, e->dt, s);
  }
}

extern "C" int etainv_maps_reset(etainv_engine_t* e, void* stream) {
  ETAINV_CHECK(e, "null engine");
  ETAINV_HIP(hipMemsetAsync(e->maps_cur, 0, e->maps_cur_bytes, (hipStream_t)stream));
  return 0;
}

extern "C" int etainv_maps_configure(etainv_engine_t* e, int res_div, void* stream) {
  ETAINV_CHECK(e, "null engine");
  ETAINV_CHECK(res_div == 2 || res_div == 4 || res_div == 8, "res_div must be 2, 4 or 8 (the (L/2)^2, (L/4)^2 cross layers or the mid block's (L/8)^2 layer)");
  ETAINV_CHECK(e->L % res_div == 0, "latent size not divisible by res_div");
  const int layers = res_div == 8 ? 1 : 5;
  const size_t res = e->L / res_div;
  const size_t bytes = (size_t)layers * e->max_img * 2 * etainv_engine::kHeads * res * res * 77 * 4;
  if (bytes <= e->maps_bytes) {
    e->maps_cur = e->maps_acc;
  } else {
    if (bytes > e->maps_alt_bytes) {   // first use of the (L/2)^2 store: its own allocation (host work, outside any capture)
      ETAINV_HIP(hipStreamSynchronize((hipStream_t)stream));
      e->maps_cur = e->maps_acc;           // (a failure below leaves the engine on its default store, never on a freed one)
      e->maps_cur_bytes = e->maps_bytes;
      e->map_div = 4;
      e->map_layers = 5;
      float* old = e->maps_alt;
      e->maps_alt = nullptr;
      e->maps_alt_bytes = 0;
      if (old) ETAINV_HIP(hipFree(old));
      ETAINV_HIP(hipMalloc(reinterpret_cast<void**>(&e->maps_alt), bytes));
      e->maps_alt_bytes = bytes;
    }
    e->maps_cur = e->maps_alt;
  }
  e->maps_cur_bytes = bytes;
  e->map_div = res_div;
  e->map_layers = layers;
  ETAINV_HIP(hipMemsetAsync(e->maps_cur, 0, e->maps_cur_bytes, (hipStream_t)stream));
  return 0;
}

extern "C" int etainv_maps_word_maps_ex(etainv_engine_t* e, int n_img, const int32_t* tokens, int n_tok, int steps_done, int row_sel, unsigned layer_mask,
                                        float* out, int accumulate, float scale, void* stream) {
  ETAINV_CHECK(e && n_img >= 1 && n_img <= e->max_img && (row_sel == 0 || row_sel == 1), "bad arguments");
  ETAINV_CHECK((layer_mask & ((1u << e->map_layers) - 1u)) != 0, "layer_mask selects none of the stored layers");
  return launch_word_maps(e->maps_cur, e->map_layers, e->max_img, 2, row_sel, etainv_engine::kHeads, e->L / e->map_div, e->L, n_img, tokens, n_tok,
                          steps_done, out, accumulate, scale, (hipStream_t)stream, layer_mask);
}

extern "C" int etainv_maps_word_maps(etainv_engine_t* e, int n_img, const int32_t* tokens, int n_tok, int steps_done, float* out,
                                     int accumulate, float scale, void* stream) {
  ETAINV_CHECK(e && n_img >= 1 && n_img <= e->max_img, "bad arguments");
  return launch_word_maps(e->maps_cur, e->map_layers, e->max_img, 2, 0, etainv_engine::kHeads, e->L / e->map_div, e->L, n_img, tokens, n_tok, steps_done, out,
                          accumulate, scale, (hipStream_t)stream);
}

extern "C" int etainv_maps_word_maps_role(etainv_engine_t* e, int n_img, const int32_t* tokens, int n_tok, int steps_done, int row_sel,
                                          float* out, int accumulate, float scale, void* stream) {
  ETAINV_CHECK(e && n_img >= 1 && n_img <= e->max_img && (row_sel == 0 || row_sel == 1), "bad arguments");
  return launch_word_maps(e->maps_cur, e->map_layers, e->max_img, 2, row_sel, etainv_engine::kHeads, e->L / e->map_div, e->L, n_img, tokens, n_tok, steps_done, out,
                          accumulate, scale, (hipStream_t)stream);
}

extern "C" int etainv_local_blend(etainv_engine_t* e, float* x, int n_img, const float* blend_alpha, float thres, void* stream) {
  ETAINV_CHECK(e && n_img >= 1 && n_img <= e->max_img, "bad arguments");
  ETAINV_CHECK(e->map_div == 4, "LocalBlend reads the five (L/4)^2 cross layers (ptp.py:37-39): etainv_maps_configure(e, 4, ...) first");
  return launch_local_blend(e->maps_acc, 5, e->max_img, etainv_engine::kHeads, e->L / 4, e->L, x, n_img, blend_alpha, thres,
                            (hipStream_t)stream);
}

extern "C" int etainv_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}
extern "C" int etainv_prof_reset(void) {
  g_prof.clear();
  g_prof_used = 0;
  return 0;
}
/* Sum of event-timed durations (ms), work (FLOPs or bytes) and launch count of one kernel class since the last reset.
 * Synchronises the device (diagnostic call, never used inside the loops). */
extern "C" int etainv_prof_read(int cls, double* ms, double* work, int64_t* launches) {
  ETAINV_CHECK(cls >= 0 && cls < PROF_NCLASS && ms && work && launches, "bad arguments");
  ETAINV_HIP(hipDeviceSynchronize());
  double t = 0, w = 0;
  int64_t n = 0;
  for (auto& r : g_prof)
    if (r.cls == cls) {
      float f = 0.f;
      ETAINV_HIP(hipEventElapsedTime(&f, r.a, r.b));
      t += f;
      w += r.work;
      ++n;
    }
  *ms = t;
  *work = w;
  *launches = n;
  return 0;
}
extern "C" int etainv_prof_records(int cls, double* ms, double* work, int64_t cap, int64_t* launches) {
  ETAINV_CHECK(cls >= 0 && cls < PROF_NCLASS && ms && work && launches && cap >= 0, "bad arguments");
  ETAINV_HIP(hipDeviceSynchronize());
  int64_t n = 0;
  for (auto& r : g_prof)
    if (r.cls == cls) {
      if (n < cap) {
        float f = 0.f;
        ETAINV_HIP(hipEventElapsedTime(&f, r.a, r.b));
        ms[n] = f;
        work[n] = r.work;
      }
      ++n;
    }
  *launches = n;
  return 0;
}

/* etainv_prof_records + the algorithmic HBM bytes of every launch (0 for classes that record none): joins a per-launch PMC traffic pass
 * (tools/pmc_per_launch.py) with what each launch had to move at minimum */
extern "C" int etainv_prof_records_ex(int cls, double* ms, double* work, double* bytes, int64_t cap, int64_t* launches) {
  ETAINV_CHECK(cls >= 0 && cls < PROF_NCLASS && ms && work && bytes && launches && cap >= 0, "bad arguments");
  ETAINV_HIP(hipDeviceSynchronize());
  int64_t n = 0;
  for (auto& r : g_prof)
    if (r.cls == cls) {
      if (n < cap) {
        float f = 0.f;
        ETAINV_HIP(hipEventElapsedTime(&f, r.a, r.b));
        ms[n] = f;
        work[n] = r.work;
        bytes[n] = r.bytes;
      }
      ++n;
    }
  *launches = n;
  return 0;
}

/* Roofline split of one class: launches whose arithmetic intensity work / bytes is at least `ridge` (FLOP per byte; MFMA peak / HBM peak) are
 * MFMA-bound, the rest HBM-bound.  out[0..2] = ms, FLOPs, bytes of the MFMA-bound launches; out[3..5] = the same of the HBM-bound ones. */
extern "C" int etainv_prof_split(int cls, double ridge, double* out6, int64_t* launches2) {
  ETAINV_CHECK(cls >= 0 && cls < PROF_NCLASS && out6 && launches2, "bad arguments");
  ETAINV_HIP(hipDeviceSynchronize());
  for (int k = 0; k < 6; ++k) out6[k] = 0.0;
  launches2[0] = launches2[1] = 0;
  for (auto& r : g_prof)
    if (r.cls == cls && r.bytes > 0.0) {
      float f = 0.f;
      ETAINV_HIP(hipEventElapsedTime(&f, r.a, r.b));
      const int o = (r.work / r.bytes >= ridge) ? 0 : 3;
      out6[o] += f;
      out6[o + 1] += r.work;
      out6[o + 2] += r.bytes;
      ++launches2[o / 3];
    }
  return 0;
}

// ---- per-op entry points for the parity tests
extern "C" int etainv_op_gemm(const void* a, const void* w, const void* bias, const void* residual, void* out, int m, int n, int k,
                              int geglu, int dtype, void* stream) {
  IGemmParams p;
  p.a1 = a;
  p.w = w;
  p.bias = (const float*)bias;
  p.residual = residual;
  p.out = out;
  p.M = m;
  p.N = n;
  p.c1 = k;
  p.W = m;
  p.Wo = m;
  p.geglu = geglu;
  p.rows_per_batch = m;
  return launch_igemm(p, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_gemm_ln(const void* a, const void* w_folded, const float* c_vec, const float* s_vec, const float* stat,
                                 const void* residual, void* out, float* stat_out, int* stat_p_out, int m, int n, int k, int geglu,
                                 int dtype, void* stream) {
  IGemmParams p;
  p.a1 = a;
  p.w = w_folded;
  p.bias = c_vec;
  p.residual = residual;
  p.out = out;
  p.M = m;
  p.N = n;
  p.c1 = k;
  p.W = m;
  p.Wo = m;
  p.geglu = geglu;
  p.rows_per_batch = m;
  if (stat) {
    p.ln_stat = stat;
    p.ln_s = s_vec;
  }
  p.stat_out = stat_out;
  return launch_igemm(p, dtype, (hipStream_t)stream, stat_p_out);
}

extern "C" int etainv_op_gemm_gnstat(const void* a, const void* w, const float* bias, const void* residual, void* out, float* part, int* wm_out, int m,
                                     int n, int k, int rows_per_image, int dtype, void* stream) {
  IGemmParams p;
  p.a1 = a;
  p.w = w;
  p.bias = bias;
  p.residual = residual;
  p.out = out;
  p.M = m;
  p.N = n;
  p.c1 = k;
  p.W = m;
  p.Wo = m;
  p.rows_per_batch = rows_per_image;
  p.stat_out = part;
  p.stat_kind = 1;
  return launch_igemm(p, dtype, (hipStream_t)stream, wm_out);
}

extern "C" int etainv_op_conv3x3_gnstat(const void* x_nhwc, const void* w_okkc, const float* bias, const float* rowvec, const void* residual, void* out,
                                        float* part, int* wm_out, int b, int h, int wd, int cin, int cout, int dtype, void* stream) {
  IGemmParams p;
  p.a1 = x_nhwc;
  p.w = w_okkc;
  p.bias = bias;
  p.rowvec = rowvec;
  p.rowvec_stride = cout;
  p.residual = residual;
  p.out = out;
  p.c1 = cin;
  p.H = p.Ho = h;
  p.W = p.Wo = wd;
  p.taps = 9;
  p.M = b * h * wd;
  p.N = cout;
  p.rows_per_batch = h * wd;
  p.stat_out = part;
  p.stat_kind = 1;
  return launch_igemm(p, dtype, (hipStream_t)stream, wm_out);
}

extern "C" int etainv_op_groupnorm_pre(const void* x1, const void* x2, int c1, int c2, const float* part1, int wm1, const float* part2, int wm2,
                                       const float* gamma, const float* beta, void* out, int b, int hw, int groups, float eps, int silu,
                                       float* final_stats, int dtype, void* stream) {
  return launch_groupnorm_pre(x1, x2, c1, c2, part1, wm1, part2, wm2, gamma, beta, out, b, hw, groups, eps, silu, final_stats, dtype,
                              (hipStream_t)stream);
}

extern "C" int etainv_op_ln_fold(const float* w, const float* gamma, const float* beta, const float* bias, int n, int k, int geglu, float scale,
                                 void* w_out, float* s_out, float* c_out, int dtype, void* stream) {
  const int mode = geglu ? PK_GEGLU : PK_PLAIN;
  // the bias goes through the same row permutation as the weight (into c_out, which the fold then reads and overwrites element by element)
  if (bias && launch_pack_weight(bias, c_out, n, 1, mode, 1, ETAINV_F32, (hipStream_t)stream)) return 1;
  return launch_ln_fold(w, gamma, beta, bias ? c_out : nullptr, n, k, mode, scale, w_out, s_out, c_out, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_row_stats(const void* x, float* stat, int rows, int c, float eps, int dtype, void* stream) {
  return launch_row_stats(x, stat, rows, c, eps, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_ln_finalize(const float* partials, int p, int cw, float eps, float* stat, int rows, void* stream) {
  return launch_ln_finalize(partials, p, cw, eps, stat, rows, (hipStream_t)stream);
}

extern "C" int etainv_op_conv3x3_ex(const void* x_nhwc, const void* w_okkc, const void* bias, const void* residual, void* out, int b, int h,
                                    int wd, int cin, int cout, int stride, int upsample, int pad0, int out_nchw, int out_io_dtype,
                                    int dtype, void* stream) {
  IGemmParams p;
  p.a1 = x_nhwc;
  p.w = w_okkc;
  p.bias = (const float*)bias;
  p.residual = residual;
  p.out = out;
  p.c1 = cin;
  p.H = h;
  p.W = wd;
  p.Ho = upsample ? h * 2 : (stride == 2 ? h / 2 : h);
  p.Wo = upsample ? wd * 2 : (stride == 2 ? wd / 2 : wd);
  p.stride = stride;
  p.ups = upsample;
  p.taps = 9;
  p.pad0 = pad0;
  p.out_nchw = out_nchw;
  p.out_io_dtype = out_io_dtype;
  p.M = b * p.Ho * p.Wo;
  p.N = cout;
  p.rows_per_batch = p.Ho * p.Wo;
  return launch_igemm(p, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_pack_ups4(const float* w_oc33, void* dst, int cout, int cin, int dtype, void* stream) {
  return launch_pack_ups4(w_oc33, dst, cout, cin, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_conv3x3(const void* x_nhwc, const void* x2_nhwc, int c1, int c2, const void* w_okkc, const void* bias,
                                 const float* rowvec, const void* residual, void* out, int b, int h, int wd, int cout, int stride,
                                 int upsample, int taps, int dtype, void* stream) {
  IGemmParams p;
  p.a1 = x_nhwc;
  p.a2 = x2_nhwc;
  p.w = w_okkc;
  p.bias = (const float*)bias;
  p.rowvec = rowvec;
  p.rowvec_stride = cout;
  p.residual = residual;
  p.out = out;
  p.c1 = c1;
  p.c2 = c2;
  p.H = h;
  p.W = wd;
  p.Ho = upsample ? h * 2 : (stride == 2 ? h / 2 : h);
  p.Wo = upsample ? wd * 2 : (stride == 2 ? wd / 2 : wd);
  p.stride = stride;
  p.ups = upsample;
  p.taps = taps;
  p.M = b * p.Ho * p.Wo;
  p.N = cout;
  p.rows_per_batch = p.Ho * p.Wo;
  return launch_igemm(p, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_groupnorm(const void* x_nhwc, const void* x2_nhwc, int c1, int c2, const float* gamma, const float* beta,
                                   void* out, int b, int hw, int groups, float eps, int silu, float* scratch, int dtype, void* stream) {
  return launch_groupnorm(x_nhwc, x2_nhwc, c1, c2, gamma, beta, out, b, hw, groups, eps, silu, scratch, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_layernorm(const void* x, const float* gamma, const float* beta, void* out, int rows, int c, float eps, int dtype,
                                   void* stream) {
  return launch_layernorm(x, gamma, beta, out, rows, c, eps, dtype, (hipStream_t)stream);
}

#ifdef ETAINV_EXPERIMENTS
extern "C" int etainv_experiments_built = 1;   // data symbol (not part of the ABI of include/etainv.h): tests of the opt-in experiments look for it
#endif

extern "C" int etainv_op_self_attention(const void* qkv, void* out, int b, int n, int heads, int d, int mode, int n_img, int dtype,
                                        void* stream) {
  return launch_self_attention_mode(qkv, out, b, n, heads, d, mode, n_img, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_cross_attention(const void* q, const void* kv, void* out, int b, int n, int heads, int d, int n_ctx,
                                         const etainv_attn_ctrl* ctrl, int map_layer, int n_img_cap, float* maps_acc, int dtype,
                                         void* stream) {
  CrossParams cp;
  cp.N = n;
  cp.heads = heads;
  cp.n_ctx = n_ctx;
  cp.scale_log2 = (1.0f / std::sqrt((float)d)) * 1.4426950408889634f;
  cp.rows = b;
  cp.n_img_cap = n_img_cap;
  cp.maps_acc = maps_acc;
  if (ctrl && (ctrl->mode == ETAINV_ATTN_PTP || ctrl->mode == ETAINV_ATTN_STORE)) {
    cp.n_img = ctrl->n_img;
    cp.layout = ctrl->mode == ETAINV_ATTN_PTP ? 2 : 1;
    cp.first_row = ctrl->mode == ETAINV_ATTN_PTP ? ctrl->first_row : 0;
    if (ctrl->mode == ETAINV_ATTN_PTP && ctrl->cross_alpha && (ctrl->mapper || ctrl->replace_mat)) {
      cp.edit = 1;
      cp.mapper = ctrl->mapper;
      cp.alphas = ctrl->alphas;
      cp.replace_mat = ctrl->replace_mat;
      cp.equalizer = ctrl->equalizer;
      cp.cross_alpha = ctrl->cross_alpha;
    }
    if (ctrl->store_maps && maps_acc) cp.map_layer = map_layer;
  }
  return launch_cross_attention_p(q, kv, out, b, d, cp, dtype, (hipStream_t)stream);
}

extern "C" int etainv_op_word_maps(const float* maps_acc, int n_layers, int n_img_cap, int heads, int res, int L, int n_img,
                                   const int32_t* tokens, int n_tok, int steps_done, float* out, int accumulate, float scale,
                                   void* stream) {
  return launch_word_maps(maps_acc, n_layers, n_img_cap, 2, 0, heads, res, L, n_img, tokens, n_tok, steps_done, out, accumulate, scale,
                          (hipStream_t)stream);
}

extern "C" int etainv_op_local_blend(const float* maps_acc, int n_layers, int n_img_cap, int heads, int res, int L, float* x, int n_img,
                                     const float* blend_alpha, float thres, void* stream) {
  return launch_local_blend(maps_acc, n_layers, n_img_cap, heads, res, L, x, n_img, blend_alpha, thres, (hipStream_t)stream);
}
