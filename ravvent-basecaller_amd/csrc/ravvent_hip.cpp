// libravvent_hip.so -- host side of the C-ABI declared in include/ravvent_hip.h.
// Owns device memory, the stream, the decode hipGraph cache and the launch sequence that
// stands behind Basecaller.beam_search_prediction / greedy_search_prediction
// (/root/reference/basecaller.py:296-330).  No CPU compute path exists in this library.
#include "../../include/ravvent_hip.h"
#include "common.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace {

std::string g_create_error;

struct LstmW { const float *W, *U, *b; };

struct ProfEntry { double ms = 0; int64_t n = 0; };

constexpr int RV_MAX_ASYNC = 16;

struct GraphKey {
  int B, W, Tm, L, greedy, taps, split;
  bool operator<(const GraphKey& o) const {
    return std::tie(B, W, Tm, L, greedy, taps, split) < std::tie(o.B, o.W, o.Tm, o.L, o.greedy, o.taps, o.split);
  }
};

// whole-slab graphs: everything of a call that ends up in a frozen kernel argument or decides which kernels run
struct SlabKey {
  int B, T_r, T_e, W, L, flags;             // flags: greedy | calls << 1 | host inputs << 2 | host outputs << 3
  uint64_t lut;                             // the fused post-processing's letter table travels by value in a kernel argument
  bool operator<(const SlabKey& o) const {
    return std::tie(B, T_r, T_e, W, L, flags, lut) < std::tie(o.B, o.T_r, o.T_e, o.W, o.L, o.flags, o.lut);
  }
};

}  // namespace

struct RvContext {
  RvConfig cfg{};
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<void*> allocs;

  // weights
  float* d_w = nullptr;
  size_t n_w = 0;
  bool loaded = false;
  std::vector<std::vector<LstmW>> enc[2];   // [enc][layer][dir]
  std::vector<LstmW> dec;                   // stacked decoder cells
  float* d_WmemT = nullptr;                 // derived: W_mem^T [128][256]
  int opt_split = 1;                        // concurrent decode sub-slabs (1..4); measured neutral at B=256
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  int opt_att_nt = 0;
  int opt_side_ev = 1;                      // event chain on a side stream under the raw in-projection GEMM
  int opt_persist = 1;                      // whole beam-search loop in one launch, attention memory register-resident
  int* d_chunk_steps = nullptr;
  int lpersist = 0;
  int opt_flash = 1;                        // single-pass Luong attend (two-pass when 0 / Bahdanau)
  int lflash = 0, lkeys = 0, lsplit = 1;
  float* d_Up = nullptr;                    // derived: recurrent kernels in the recurrence kernels' register order, [enc][layer][dir][65536]
  float* d_Wp = nullptr;                    // derived: input kernels of encoder layers >= 1 as MFMA B fragments, [enc][layer-1][dir][131072]
  uint16_t* d_Wsb = nullptr;                // derived: the same kernels cut into three bf16 parts, [enc][layer-1][dir][32 tiles][3 parts][8 k-steps][64 lanes][8]
  uint16_t* d_Wh = nullptr;                 // derived: ... as two f16 parts of the column-scaled kernels + 512 column factors, [enc][layer-1][dir][RV_WH_SLOT]
  int opt_split_proj = 2;                   // fused projection on split-bf16 MFMAs (six part products, f32-equivalent); 0 = f32 MFMAs
  int opt_mx_att = 1;                       // persistent decode (Luong, one cell): attention on the matrix pipe
  int opt_mx_cell = 1;                      // ... and its cell product [ctx' | h] . Wcat2 as well (needs opt_mx_att)
  float mx_cdescale = 1.f;                  // 1 / the power-of-two scale of the Wc16 image (set by rv_load_weights)
  float mx_ldescale = 1.f;                  // ... of the Wl16 image
  uint16_t* d_Wl16 = nullptr;               // derived (one decoder cell): [W_fc ; A_h W_fc] as MFMA B fragments (DecState::Wl16)
  uint16_t* d_Wq16 = nullptr;               // derived (Bahdanau, one decoder cell): W_q as MFMA B fragments (DecState::Wq16)
  float mx_qdescale = 1.f;
  uint16_t* d_W1c16 = nullptr;              // derived (two decoder cells): [W_1 | U_1] as MFMA B fragments (DecState::W1c16)
  float mx_c1descale = 1.f;
  float mx_kscale = 1.f, mx_uscale = 1.f;   // powers of two from the bounds of [keys | U'] = enc_out . Wmp (set by rv_load_weights)
  uint16_t* d_Ua = nullptr;                 // derived: recurrent kernels as MFMA A fragments (two f16 parts) + row factors, [enc][layer][dir][RV_UA_SLOT]
  uint16_t* d_Wx16 = nullptr;               // derived: input kernels of encoder layers >= 1, both directions, as the split GEMM's B slabs, [enc][layer-1][RV_WX16_SLOT]
  float* d_bx2 = nullptr;                   // derived: [b_fwd | b_bwd] of those layers, [enc][layer-1][1024]
  int opt_wide = 1;                         // matrix-pipe recurrence, 16 chunks per workgroup: 1 on (default: every call runs the same kernels whatever its slab
                                            // size, so results do not depend on how a read is cut into slabs / shards), 0 packed-FMA kernels, -1 per-call choice
  int lwide = 0;
  int opt_lane_inproj = 1;                  // the event encoder's layer-0 input projection inside its matrix-pipe recurrence (0: k_inproj_small + pre-projected inputs)
  int lrows8 = 0;                           // this call's matrix-pipe recurrences take eight chunks per workgroup (the latency form of lstm_mx.hip)
  int opt_tail_wave = 1;                    // layer 0: cell update on a ninth wave, two row groups half a step apart
  int opt_fuse = 1;                         // layers >= 1: input projection inside the recurrence kernel (MFMA waves)
  int dbg_role = 0;                         // timing probe (RV_DBG_ROLE): 1 = no projection math, 2 = no recurrence math
  float* d_Wmp = nullptr;                   // derived: [W_mem | A_c] [256][256] (A_c = W_att rows 128..383): projection of the attention memory for the persistent decode
  float* d_Wcat2 = nullptr;                 // derived (one decoder cell): [W_a ; U + A_h W_a] [256][512]
  float* d_Nh = nullptr;                    // derived (one decoder cell): A_h W_fc [128][V]
  uint16_t* d_Wc16 = nullptr;               // derived (one decoder cell): Wcat2 as two f16 parts in MFMA B-fragment order (DecState::Wc16)
  uint16_t* d_Wmp16 = nullptr;              // derived: Wmp as two f16 parts in MFMA fragment order + column factors (launch_gemm_mem_split)
  float* mem2 = nullptr;                    // [B,Tm,256] = enc_out . Wmp: keys | attention-layer image of the values
  float* d_WcatT = nullptr;                 // derived: ([W_dec[V:] ; U_dec])^T, [512][256]
  const float *W_mem = nullptr, *W_q = nullptr, *v_att = nullptr, *W_att = nullptr, *W_fc = nullptr, *b_fc = nullptr;

  // encoder buffers
  float *d_raw = nullptr, *d_ev = nullptr;
  uint8_t* mask = nullptr;
  float* act[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [enc][pingpong]
  float* xw[2] = {nullptr, nullptr};   // per encoder (the two encoders run on concurrent streams)
  float* st[2][2][4] = {};             // [enc][set][h_f, c_f, h_b, c_b]
  float *enc_out = nullptr, *keys = nullptr;
  // decoder buffers
  DecState dec_st{};
  float* step_align = nullptr; size_t step_align_cap = 0;
  int32_t* out_tokens = nullptr;
  float* out2 = nullptr;
  // pinned host staging for the host-buffer entry points (pageable hipMemcpy is synchronous and slow)
  float *pin_raw = nullptr, *pin_ev = nullptr, *pin_out2 = nullptr;
  int32_t* pin_tok = nullptr;
  int* pin_S = nullptr;
  uint8_t *d_bases = nullptr, *pin_bases = nullptr;
  float *d_probs = nullptr, *pin_probs = nullptr;
  int *d_clen = nullptr, *pin_clen = nullptr;

  int opt_taps = 0, opt_graph = 1, opt_profile = 0;
  long long* rec_ts = nullptr;              // diagnostic: per-wave cycle sums of the raw layer-0 recurrence (RV_REC_STAMPS=1) or of its fused layer 1 (=2)
  int rec_ts_layer = 0;
  int opt_ptaps = 0, lptaps = 0;            // persist_taps: per-step logits of the persistent decode (debug)
  std::map<std::string, ProfEntry> prof;
  struct Pending { std::string name; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;
  std::map<GraphKey, hipGraphExec_t> graphs;
  // one hipGraph per slab context and call shape: the whole slab (ten launches on the default path) replays as ONE hipGraphLaunch.
  // Every kernel argument is frozen at capture; the addresses that change from call to call (caller inputs and outputs) reach the
  // kernels through a 4-entry table in mapped pinned memory (common.h: RV_PTAB_*), rewritten by the host before each replay
  struct SlabGraph { hipGraphExec_t exec = nullptr; int lflash = 0, lkeys = 0, lpersist = 0, lsplit = 1, lW = 0; };
  std::map<SlabKey, SlabGraph> slab_graphs;
  const void** d_ptab = nullptr;
  const void** pin_ptab = nullptr;
  int opt_slab_graph = 0;                   // option "slab_graph": off by default -- measured (tools/graph_ab.py, C3, depth 10): a submit costs the host
                                            // 22-27 us instead of 43-48, but the replayed slabs stream 1.5-3 % SLOWER (354-361 k against 361-370 k chunks/s
                                            // settled; the runtime's per-node barrier packets), and the host is not what bounds the stream
  int opt_gen = 0;                          // (root) bumped by every rv_set_option / rv_load_weights: frozen arguments may have changed
  int graph_gen = 0;                        // the opt_gen this context's slab graphs were captured under

  // last call
  int lB = 0, lW = 0, lTm = 0, lL = 0, lS = 0, lgreedy = 0, ltaps = 0;

  // asynchronous calls (rv_beam_search_submit* / rv_beam_search_collect*): the handle owns `kids` more slab contexts -- own stream,
  // own buffers, the PARENT's weights and derived images -- so that several slabs are in flight on the GPU at once
  RvContext* parent = nullptr;
  std::vector<RvContext*> kids;
  int opt_async_depth = 2;                  // contexts the asynchronous entry points rotate through (1..RV_MAX_ASYNC)
  int inflight_hint = 1;                    // slabs the caller keeps in flight (1 for the synchronous entry points): sizes the workgroups
  int next_slot = 0, generation = 0;
  struct PendingCall { bool busy = false, trivial = false, greedy = false, dev_out = false, calls = false; int B = 0, steps = 0, V = 0, Wd = 0, ticket = -1; } pend;
};

namespace {

int fail(RvContext* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(h, RV_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <class T>
int dalloc(RvContext* h, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) return fail(h, RV_ENOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
  h->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return RV_OK;
}

#define RV_WH_SLOT ((size_t)2 * RV_E * RV_G + 2 * RV_G)   // uint16 per (encoder, layer, direction) of d_Wh
// bf16 with round-to-nearest-even, and back (finite inputs: the weights)
inline uint16_t bf16_rne(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_value(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float x; memcpy(&x, &u, 4);
  return x;
}

size_t weight_count(const RvConfig& c) {
  const size_t u = c.enc_units, d = c.dec_units, V = c.vocab;
  size_t n = 0;
  for (int e = 0; e < 2; ++e)
    for (int l = 0; l < c.enc_depth; ++l) {
      const size_t F = l > 0 ? 2 * u : (e == 0 ? 1 : 5);
      n += 2 * (F * 4 * u + u * 4 * u + 4 * u);
    }
  for (int k = 0; k < c.dec_depth; ++k) {
    const size_t fin = k == 0 ? V + d : d;
    n += fin * 4 * d + d * 4 * d + 4 * d;
  }
  n += 2 * u * d + d * d + d + (d + 2 * u) * d + d * V + V;
  return n;
}

void bind_weights(RvContext* h) {
  const RvConfig& c = h->cfg;
  const size_t u = c.enc_units, d = c.dec_units, V = c.vocab;
  const float* p = h->d_w;
  for (int e = 0; e < 2; ++e) {
    h->enc[e].assign(c.enc_depth, std::vector<LstmW>(2));
    for (int l = 0; l < c.enc_depth; ++l) {
      const size_t F = l > 0 ? 2 * u : (e == 0 ? 1 : 5);
      for (int dr = 0; dr < 2; ++dr) {
        LstmW& w = h->enc[e][l][dr];
        w.W = p; p += F * 4 * u;
        w.U = p; p += u * 4 * u;
        w.b = p; p += 4 * u;
      }
    }
  }
  h->dec.assign(c.dec_depth, LstmW{});
  for (int k = 0; k < c.dec_depth; ++k) {
    h->dec[k].W = p; p += (k == 0 ? V + d : d) * 4 * d;
    h->dec[k].U = p; p += d * 4 * d;
    h->dec[k].b = p; p += 4 * d;
  }
  h->W_mem = p; p += 2 * u * d;
  h->W_q = p; p += d * d;
  h->v_att = p; p += d;
  h->W_att = p; p += (d + 2 * u) * d;
  h->W_fc = p; p += d * V;
  h->b_fc = p; p += V;
}

// ---- profiling: bracket a launch with events on the launch stream
struct Scope {
  RvContext* h; const char* name; hipEvent_t a = nullptr, b = nullptr; bool on; hipStream_t st;
  // profile 3: only the decode launches (the dominant kernel) are bracketed, so that the timed region of bench.py
  // carries two events per slab instead of two per launch (the per-launch events cost 2.6 % of a C3 slab)
  Scope(RvContext* h_, const char* n, hipStream_t st_ = nullptr, bool decode = false)
      : h(h_), name(n), on(h_->opt_profile != 0 && (h_->opt_profile != 3 || decode)), st(st_ ? st_ : h_->stream) {
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); }
      else hipEventCreate(&e);
      return e;
    };
    a = get(); b = get();
    hipEventRecord(a, st);
  }
  ~Scope() {
    if (!on) return;
    hipEventRecord(b, st);
    h->pending.push_back({name, a, b});
  }
};

void drain_profile(RvContext* h) {
  RvContext* root = h->parent ? h->parent : h;
  for (auto& p : h->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ProfEntry& e = root->prof[p.name];
      e.ms += ms; e.n += 1;
    }
    h->ev_pool.push_back(p.a); h->ev_pool.push_back(p.b);
  }
  h->pending.clear();
}

// Side stream g of a context, created on first use.  Highest priority: their few workgroups should take CU slots as soon as they free up.
hipStream_t side_stream(RvContext* h, int g) {
  if (!h->side[g]) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&h->side[g], hipStreamNonBlocking, hi) != hipSuccess) { h->side[g] = nullptr; return h->stream; }   // (degrades to in-order)
  }
  return h->side[g];
}

int pick_rows_per_block(int B) {
  // one direction of BT chunks per workgroup, 2 directions: aim at <= 256 workgroups (1 per CU)
  int bt = (2 * B + 255) / 256;
  int r = 1;
  while (r < bt && r < 8) r <<= 1;
  return r;
}

// Encoder.call for one encoder (basecaller.py:48-59) writing into enc_out at time offset t_off.
// Does this slab run its encoder recurrences on the matrix pipe (lstm_mx.hip: 16 chunks of one direction per workgroup)?
bool wide_recurrence(const RvContext* h, int B, int T_r) {
  if (h->opt_wide == 0 || !h->opt_fuse) return false;
  if (h->cfg.mode != RV_MODE_EVENT && !lstm_rec_mx_window_fits(T_r)) return false;
  if (h->opt_wide > 0) return true;
  // Per-call choice (-1).  The packed-FMA kernels cost per chunk, the matrix forms per workgroup.  Synchronous calls at T = 300 + 30
  // (tools/rows8_ab.py, k chunks/s; FMA / matrix pipe with 8 / with 16 chunks per workgroup): 64 chunks 75 / 64 / 53; 128: 146 / 124 / 102;
  // 192: 159 / 181 / 150; 256: 203 / 224 / 189; 512: 208 / 298 / 266; 1,024: 212 / 342 / 341 -- FMA below 160 chunks in flight, the
  // eight-chunk form up to 512, sixteen above (less CU-time per chunk: what a stream of slabs wants).
  return (long long)B * std::max(h->inflight_hint, 1) >= 160;
}

// ... and with eight chunks per workgroup (two MFMAs per tile and k-step, half the cell update per lane: a shorter step on twice the
// workgroups)?  wide_recurrence = 2 always; = -1 (per-call choice) when the slabs in flight leave the chip room for it.
bool rows8_wanted(const RvContext* h, int B) {
  if (h->opt_wide == 2) return true;
  const long long n = (long long)B * std::max(h->inflight_hint, 1);
  return h->opt_wide < 0 && n >= 160 && n <= 512;
}

void run_encoder(RvContext* h, int e, const float* x, int F, int B, int T, int Tm, int t_off, hipStream_t s,
                 int l_begin = 0, int l_end = 1 << 30, const void* const* ptab = nullptr) {
  const int depth = h->cfg.enc_depth;
  const int bt = pick_rows_per_block(B);
  for (int l = std::max(l_begin, 0); l < std::min(depth, l_end); ++l) {
    const bool last = l == depth - 1;
    float* out = last ? h->enc_out : h->act[e][l & 1];
    RecArgs a{};
    a.B = B; a.T = T;
    a.out = out; a.out_T = last ? Tm : T; a.out_t0 = last ? t_off : 0;
    const int rd = (l & 1) ^ 1, wr = l & 1;     // state sets: layer l reads set rd, writes set wr
    for (int dr = 0; dr < 2; ++dr) {
      const LstmW& w = h->enc[e][l][dr];
      a.U[dr] = w.U;
      a.Up[dr] = h->d_Up + ((size_t)(e * depth + l) * 2 + dr) * RV_U * RV_G;
      a.Ua[dr] = h->d_Ua + ((size_t)(e * depth + l) * 2 + dr) * RV_UA_SLOT;
      a.h0[dr] = l > 0 ? h->st[e][rd][2 * dr] : nullptr;
      a.c0[dr] = l > 0 ? h->st[e][rd][2 * dr + 1] : nullptr;
      a.hT[dr] = h->st[e][wr][2 * dr];
      a.cT[dr] = h->st[e][wr][2 * dr + 1];
    }
    if (h->lwide) {
      // matrix-pipe recurrence.  Raw layer 0 folds its one-feature input projection in; every other layer reads pre-projected
      // inputs xw [B,T,2,512]: event layer 0 from a small elementwise kernel, layers >= 1 from the split-f16 GEMM (both directions
      // and their biases in one launch, A read once).
      if (l == 0 && (F == 1 || (F == 5 && h->opt_lane_inproj && lstm_rec_mx_window_fits(T, 5)))) {
        // layer 0 takes x . W + b in the lane, from LDS windows of its chunks' inputs: no projection launch, no xw round trip (round 4: the
        // event encoder's five features too -- the same fused multiply-adds in the same order as k_inproj_small: identical bits)
        a.x = x;
        for (int dr = 0; dr < 2; ++dr) { a.W[dr] = h->enc[e][0][dr].W; a.bias[dr] = h->enc[e][0][dr].b; }
        a.mask = h->mask; a.mask_T = Tm; a.mask_t0 = t_off; a.pad = h->cfg.padding_value;   // the layer-0 kernels leave the input mask too
        a.ptab = ptab;
        a.dbg_ts = (e == 0 && h->rec_ts_layer == 0) ? h->rec_ts : nullptr;     // (RV_REC_STAMPS=1: phase cycle sums of workgroup (0, 0))
        Scope sc(h, F == 1 ? "lstm_rec_raw_l0" : "lstm_rec_event_l0", s);
        launch_lstm_rec_mx(a, F, s, h->lrows8 != 0);
        a.dbg_ts = nullptr; a.ptab = nullptr; a.mask = nullptr;
        continue;
      }
      if (l == 0) {
        Scope sc(h, e == 0 ? "inproj_raw_l0" : "inproj_event_l0", s);
        launch_inproj_small(x, B * T, F, h->enc[e][0][0].W, h->enc[e][0][0].b, h->enc[e][0][1].W, h->enc[e][0][1].b, h->xw[e],
                            h->mask, T, Tm, t_off, h->cfg.padding_value, s, ptab);
      } else {
        Scope sc(h, e == 0 ? "gemm_inproj_raw" : "gemm_inproj_event", s);
        launch_gemm_split_blocks(h->act[e][(l - 1) & 1], B * T, h->d_Wx16 + (size_t)(e * (depth - 1) + (l - 1)) * RV_WX16_SLOT, 4,
                                 h->d_bx2 + (size_t)(e * (depth - 1) + (l - 1)) * 2 * RV_G, h->xw[e], 2 * RV_G, s);
      }
      a.x = h->xw[e];
      a.dbg_ts = (e == 0 && l == 1 && h->rec_ts_layer == 1) ? h->rec_ts : nullptr;    // (RV_REC_STAMPS=2)
      Scope sc(h, l == 0 ? "lstm_rec_event_l0" : (e == 0 ? "lstm_rec_raw_l1p" : "lstm_rec_event_l1p"), s);
      launch_lstm_rec_mx(a, 0, s, h->lrows8 != 0);
      a.dbg_ts = nullptr;
      continue;
    }
    if (l == 0) {
      a.x = x;
      for (int dr = 0; dr < 2; ++dr) { a.W[dr] = h->enc[e][0][dr].W; a.bias[dr] = h->enc[e][0][dr].b; }
      Scope sc(h, e == 0 ? "lstm_rec_raw_l0" : "lstm_rec_event_l0", s);
      a.tail_wave = h->opt_tail_wave;
      a.dbg_ts = (e == 0 && h->rec_ts_layer == 0) ? h->rec_ts : nullptr;
      launch_lstm_rec(a, F, bt, s);
      a.dbg_ts = nullptr;
    } else {
      const float* in = h->act[e][(l - 1) & 1];
      if (h->opt_fuse) {   // x . W + b on the matrix pipe inside the recurrence kernel: no K0 launch, no xw tensor
        a.x = in;
        a.dbg_role = h->dbg_role;
        for (int dr = 0; dr < 2; ++dr) {
          a.Wp[dr] = h->d_Wp + ((size_t)(e * (depth - 1) + (l - 1)) * 2 + dr) * RV_E * RV_G;
          a.Wh[dr] = h->opt_split_proj == 2 ? h->d_Wh + ((size_t)(e * (depth - 1) + (l - 1)) * 2 + dr) * RV_WH_SLOT : nullptr;
          a.Wsb[dr] = h->opt_split_proj == 1 ? h->d_Wsb + ((size_t)(e * (depth - 1) + (l - 1)) * 2 + dr) * 3 * RV_E * RV_G : nullptr;
          a.bias[dr] = h->enc[e][l][dr].b;
        }
        a.dbg_ts = (e == 0 && l == 1 && h->rec_ts_layer == 1) ? h->rec_ts : nullptr;
        Scope sc(h, e == 0 ? "lstm_rec_raw_l1p" : "lstm_rec_event_l1p", s);
        launch_lstm_rec_proj(a, bt, s);
        continue;
      }
      {   // both directions in ONE launch (same A): 2x the workgroups -> less round quantisation
        GemmArgs g{};
        g.A = in; g.lda = RV_E; g.ldb = RV_G; g.ldc = 2 * RV_G;
        g.M = B * T; g.N = RV_G; g.K = RV_E;
        g.Bm = h->enc[e][l][0].W; g.bias = h->enc[e][l][0].b; g.C = h->xw[e];
        g.Bm1 = h->enc[e][l][1].W; g.bias1 = h->enc[e][l][1].b; g.C1 = h->xw[e] + RV_G;
        g.xcd_remap = 1;
        Scope sc(h, e == 0 ? "gemm_inproj_raw" : "gemm_inproj_event", s);
        launch_gemm_f32(g, false, s);
      }
      a.x = h->xw[e];
      Scope sc(h, e == 0 ? "lstm_rec_raw_l1p" : "lstm_rec_event_l1p", s);
      launch_lstm_rec(a, 0, bt, s);
    }
  }
}

void launch_decode_steps(RvContext* h, const DecState& d, hipStream_t s, bool profiled) {
  const bool flash = h->lflash != 0;
  auto cells = [&](int step, bool prof) {
    for (int k = 0; k < d.depth; ++k) {
      const float* WcatT = h->d_WcatT + (size_t)k * RV_G * RV_E;
      if (prof) { Scope sc(h, "dec_cell"); launch_dec_cell(d, k, WcatT, k == 0 ? h->dec[0].W : nullptr, h->dec[k].b, step, s); }
      else launch_dec_cell(d, k, WcatT, k == 0 ? h->dec[0].W : nullptr, h->dec[k].b, step, s);
    }
  };
  for (int step = 0; step < d.L - 1; ++step) {
    if (profiled) {
      cells(step, true);
      { Scope sc(h, "dec_attend"); launch_dec_attend(d, h->d_WmemT, flash, step, s); }
    } else {
      cells(step, false);
      launch_dec_attend(d, h->d_WmemT, flash, step, s);
    }
  }
}

struct CallsOut { const uint8_t* lut; uint8_t* bases; int32_t* lengths; float* probs; };

// Does this call take the one-launch persistent decode (if the kernel has an instantiation for it: dec_persist_supported)?
bool persist_wanted(const RvContext* h, bool greedy, int W, int Tm) {
  const RvConfig& c = h->cfg;
  const int W_eff = greedy ? 1 : W;
  return h->opt_persist && h->opt_flash && !h->opt_taps && c.dec_depth <= 2 &&
         (c.attention == RV_ATT_LUONG || (c.attention == RV_ATT_BAHDANAU && c.dec_depth == 1)) &&
         W_eff <= (c.dec_depth > 1 ? 5 : 8) && Tm <= 352;
}

int record_slab(RvContext* h, const float* xr, const float* xe, bool host_in, int B, int T_r, int T_e, int W, int L, bool greedy,
                int32_t* tk, float* o2, bool dev_out, const uint8_t* lut, const void* const* ptab);

// Everything of a call up to (not including) the wait for the stream: launches and the D2H copies into pinned staging.
// dev_out: tokens / out2 are device pointers written by the finalize kernel; else the results wait in pinned memory for finish().
// lut != nullptr: the fused post-processing (rv_beam_search_calls) instead of tokens / scores.
int enqueue(RvContext* h, const float* raw, const float* ev, bool dev_in, int B, int T_r, int T_e, int W,
            int L, bool greedy, int32_t* tokens, float* out2, bool dev_out, const uint8_t* lut) {
  if (!h) return RV_EINVAL;
  const RvConfig& c = h->cfg;
  const bool calls = lut != nullptr;
  if (h->pend.busy) return fail(h, RV_ESTATE, "this context still holds an uncollected call");
  if (!h->loaded) return fail(h, RV_ESTATE, "no weights loaded (call rv_load_weights first)");
  const bool use_raw = c.mode != RV_MODE_EVENT, use_ev = c.mode != RV_MODE_RAW;
  if (!use_raw) T_r = 0;
  if (!use_ev) T_e = 0;
  if (B < 0 || B > c.max_batch) return fail(h, RV_EINVAL, "B=%d outside [0,%d]", B, c.max_batch);
  if (use_raw && (T_r < 1 || T_r > c.max_raw_len)) return fail(h, RV_EINVAL, "T_r=%d outside [1,%d]", T_r, c.max_raw_len);
  if (use_ev && (T_e < 1 || T_e > c.max_event_len)) return fail(h, RV_EINVAL, "T_e=%d outside [1,%d]", T_e, c.max_event_len);
  if (W < 1 || W > c.max_beam || W > RV_MAX_BEAM) return fail(h, RV_EINVAL, "beam width %d outside [1,%d]", W, std::min(c.max_beam, RV_MAX_BEAM));
  if (L < 1 || L > c.max_output_len) return fail(h, RV_EINVAL, "max_output_len=%d outside [1,%d]", L, c.max_output_len);
  if ((use_raw && !raw) || (use_ev && !ev)) return fail(h, RV_EINVAL, "missing input pointer for this mode");
  if (T_r + T_e > 352) return fail(h, RV_EUNSUPPORTED, "attention memory of %d steps exceeds the 352 the decode kernel is built for", T_r + T_e);
  if (L > 64) return fail(h, RV_EUNSUPPORTED, "max_output_len %d exceeds 64", L);
  if (dev_out && B > 0 && L > 1 && (!tokens || !out2)) return fail(h, RV_EINVAL, "null output pointer");
  HIPCHK(h, hipSetDevice(c.device));
  h->lB = B; h->lW = W; h->lL = L; h->lS = 0; h->lgreedy = greedy; h->lTm = T_r + T_e; h->ltaps = h->opt_taps; h->lptaps = h->opt_ptaps;
  h->pend = RvContext::PendingCall{};
  h->pend.busy = true; h->pend.greedy = greedy; h->pend.dev_out = dev_out; h->pend.calls = calls;
  h->pend.B = B; h->pend.steps = std::max(L - 1, 0); h->pend.V = c.vocab; h->pend.Wd = greedy ? 1 : W;
  if (B == 0 || L <= 1) { h->pend.trivial = true; return RV_OK; }

  hipStream_t s = h->stream;
  const float *xr = raw, *xe = ev;
  if (!dev_in) {   // host -> pinned staging (CPU memcpy); the H2D copies are part of the slab's stream work (record_slab)
    if (use_raw) { memcpy(h->pin_raw, raw, sizeof(float) * B * T_r); xr = h->d_raw; }
    if (use_ev) { memcpy(h->pin_ev, ev, sizeof(float) * B * T_e * 5); xe = h->d_ev; }
  }

  h->lwide = wide_recurrence(h, B, T_r) ? 1 : 0;
  h->lrows8 = h->lwide && rows8_wanted(h, B) ? 1 : 0;
  if ((!h->opt_fuse && c.enc_depth > 1) || h->lwide) {   // pre-projected tensors [max_batch, T_max, 2, 512]: allocated with the context (alloc_slab_buffers); this is a safety net
    if (use_raw && c.enc_depth > 1 && !h->xw[0]) { const int rc = dalloc(h, &h->xw[0], (size_t)c.max_batch * c.max_raw_len * 2 * RV_G); if (rc != RV_OK) return rc; }
    if (use_ev && !h->xw[1]) { const int rc = dalloc(h, &h->xw[1], (size_t)c.max_batch * c.max_event_len * 2 * RV_G); if (rc != RV_OK) return rc; }
  }
  int32_t* tk = (dev_out && tokens) ? tokens : h->out_tokens;
  float* o2 = (dev_out && out2) ? out2 : h->out2;

  // ---- one hipGraph per slab context and call shape (option "slab_graph"): the default path -- matrix-pipe recurrences + persistent
  // decode, no profiling, no taps -- replays as ONE hipGraphLaunch instead of ten launches (a submit's host time is what the GPU waits
  // for whenever the slabs in flight finish together).  All kernel arguments are frozen at capture; the caller's addresses of THIS
  // call go through the pointer table (mapped pinned memory, written here, read by the kernels).
  RvContext* root = h->parent ? h->parent : h;
  if (h->graph_gen != root->opt_gen) {          // an option or the weights changed since the capture: frozen arguments may be stale
    for (auto& kv : h->slab_graphs) hipGraphExecDestroy(kv.second.exec);
    h->slab_graphs.clear();
    h->graph_gen = root->opt_gen;
  }
  const bool graphable = root->opt_slab_graph && h->lwide && !h->lrows8 && persist_wanted(h, greedy, W, T_r + T_e) && h->opt_profile == 0 && !h->opt_taps &&
                         !h->opt_ptaps && !h->rec_ts && !h->dec_st.dbg_ts && h->d_ptab;
  if (!graphable) return record_slab(h, xr, xe, !dev_in, B, T_r, T_e, W, L, greedy, tk, o2, dev_out, lut, nullptr);
  h->pin_ptab[RV_PTAB_RAW] = xr; h->pin_ptab[RV_PTAB_EVENT] = xe; h->pin_ptab[RV_PTAB_TOKENS] = tk; h->pin_ptab[RV_PTAB_OUT2] = o2;
  SlabKey key{B, T_r, T_e, W, L, (greedy ? 1 : 0) | (calls ? 2 : 0) | (dev_in ? 0 : 4) | (dev_out ? 0 : 8), 0};
  if (calls) memcpy(&key.lut, lut, std::min<size_t>(sizeof key.lut, (size_t)c.vocab));
  auto it = h->slab_graphs.find(key);
  if (it == h->slab_graphs.end()) {
    if (h->slab_graphs.size() >= 16) {          // bound the cache (callers with ever-changing slab shapes)
      for (auto& kv : h->slab_graphs) hipGraphExecDestroy(kv.second.exec);
      h->slab_graphs.clear();
    }
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = record_slab(h, xr, xe, !dev_in, B, T_r, T_e, W, L, greedy, tk, o2, dev_out, lut, h->d_ptab);
    const hipError_t ce = hipStreamEndCapture(s, &graph);      // (always: the stream must leave capture mode whatever record_slab said)
    if (rc != RV_OK) { if (graph) hipGraphDestroy(graph); return rc; }
    HIPCHK(h, ce);
    RvContext::SlabGraph g;
    const hipError_t ie = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    HIPCHK(h, ie);
    g.lflash = h->lflash; g.lkeys = h->lkeys; g.lpersist = h->lpersist; g.lsplit = h->lsplit; g.lW = h->lW;
    it = h->slab_graphs.emplace(key, g).first;
  }
  const RvContext::SlabGraph& g = it->second;
  h->lflash = g.lflash; h->lkeys = g.lkeys; h->lpersist = g.lpersist; h->lsplit = g.lsplit; h->lW = g.lW;
  HIPCHK(h, hipGraphLaunch(g.exec, s));
  return RV_OK;
}

// The stream work of one slab, in order: input copies, encoders, memory set-up, decode, finalize, output copies.  `ptab` non-null =
// the call is being captured into a slab graph: kernels that touch caller memory read its address from the table.
int record_slab(RvContext* h, const float* xr, const float* xe, bool host_in, int B, int T_r, int T_e, int W, int L, bool greedy,
                int32_t* tk, float* o2, bool dev_out, const uint8_t* lut, const void* const* ptab) {
  const RvConfig& c = h->cfg;
  const bool calls = lut != nullptr;
  const bool use_raw = c.mode != RV_MODE_EVENT, use_ev = c.mode != RV_MODE_RAW;
  hipStream_t s = h->stream;
  const int Tm = T_r + T_e, V = c.vocab, steps = L - 1;
  if (host_in) {   // pinned staging -> device
    if (use_raw) HIPCHK(h, hipMemcpyAsync(h->d_raw, h->pin_raw, sizeof(float) * B * T_r, hipMemcpyHostToDevice, s));
    if (use_ev) HIPCHK(h, hipMemcpyAsync(h->d_ev, h->pin_ev, sizeof(float) * B * T_e * 5, hipMemcpyHostToDevice, s));
  }
  // ---- _encode_input (basecaller.py:395-416)
  // The mask is first read by the memory set-up / the decode: it runs on a side stream beside the encoders instead of in front
  // of them (profiling modes keep it on the main stream so that its events pair up).
  // (with several slabs in flight the other slabs fill the chip: one stream per context then, no fork / join)
  // The matrix-pipe layer-0 kernels write it themselves (they read every input sample anyway): no launch at all then.
  const bool mask_aside = !h->lwide && (h->opt_profile == 0 || h->opt_profile == 3) && h->inflight_hint <= 1 && !h->parent;
  if (mask_aside) {
    hipStream_t sm = side_stream(h, 2);
    HIPCHK(h, hipEventRecord(h->ev_fork, s));
    HIPCHK(h, hipStreamWaitEvent(sm, h->ev_fork, 0));     // inputs (H2D copies on s) are in place
    launch_input_mask(xr, xe, B, T_r, T_e, c.padding_value, h->mask, sm);
    HIPCHK(h, hipEventRecord(h->ev_join[2], sm));
  } else if (!h->lwide) {
    Scope sc(h, "input_mask"); launch_input_mask(xr, xe, B, T_r, T_e, c.padding_value, h->mask, s);
  }
  // The two encoders are independent until the time-axis concat (basecaller.py:400-405).
  const bool side_ev = use_raw && use_ev && h->opt_side_ev && c.enc_depth > 1 && !h->opt_fuse;   // (nothing MFMA-bound to hide under once the projection is fused)
  if (side_ev) {
    // raw layer 0 alone (every recurrence workgroup needs a whole CU); then the short event chain on a side stream
    // UNDER the raw encoder's input-projection GEMM: a recurrence workgroup (2 x 168 VGPRs per SIMD, VALU-bound) and a
    // GEMM workgroup (144 registers, MFMA-bound) fit on one CU together.
    hipStream_t sev = side_stream(h, 0);
    run_encoder(h, 0, xr, 1, B, T_r, Tm, 0, s, 0, 1);
    HIPCHK(h, hipEventRecord(h->ev_fork, s));
    HIPCHK(h, hipStreamWaitEvent(sev, h->ev_fork, 0));
    run_encoder(h, 1, xe, 5, B, T_e, Tm, T_r, sev);
    run_encoder(h, 0, xr, 1, B, T_r, Tm, 0, s, 1);
    HIPCHK(h, hipEventRecord(h->ev_join[0], sev));
    HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[0], 0));
  } else if (h->lwide && use_raw && use_ev && h->inflight_hint <= 1 && !ptab && h->opt_profile == 0 && h->opt_side_ev) {
    // One isolated slab on the matrix-pipe form: every encoder launch is 2 x ceil(B / 16) workgroups (32 of the chip's 256 CUs at B = 256), so
    // the event encoder's chain (four launches, ~0.15 ms at the C3 shape) runs on a side stream BESIDE the raw encoder's instead of in front of
    // it: the synchronous call's latency drops by that much.  (With several slabs in flight the other slabs fill the chip: one stream per
    // context then, no fork / join.)  The two encoders write disjoint time ranges of enc_out and of the mask and use their own xw / act / state
    // buffers: results are identical.
    hipStream_t sev = side_stream(h, 0);
    HIPCHK(h, hipEventRecord(h->ev_fork, s));
    HIPCHK(h, hipStreamWaitEvent(sev, h->ev_fork, 0));       // inputs (H2D copies on s) are in place
    run_encoder(h, 1, xe, 5, B, T_e, Tm, T_r, sev);
    HIPCHK(h, hipEventRecord(h->ev_join[0], sev));
    run_encoder(h, 0, xr, 1, B, T_r, Tm, 0, s);
    HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[0], 0));
  } else {
    if (use_ev) run_encoder(h, 1, xe, 5, B, T_e, Tm, T_r, s, 0, 1 << 30, ptab);
    if (use_raw) run_encoder(h, 0, xr, 1, B, T_r, Tm, 0, s, 0, 1 << 30, ptab);
  }

  if (mask_aside) HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[2], 0));
  // ---- setup_memory (basecaller.py:303): keys = (enc_output * mask) . W_mem.  The single-pass Luong
  //      attend never reads keys (score_t = values_t . (W_mem q)); they are built for the two-pass
  //      kernel and for the "keys" debug tap only.
  const int W_eff = greedy ? 1 : W;
  h->lflash = (c.attention == RV_ATT_LUONG && h->opt_flash && W_eff <= 5) ? 1 : 0;   // wider beams: register budget -> two-pass
  // the persistent decode (decided below, once the decode state is set up) needs neither keys nor the per-step kernels
  const bool persist_ok = persist_wanted(h, greedy, W, Tm);
  h->lkeys = ((!h->lflash && !persist_ok) || h->opt_taps) ? 1 : 0;
  if (h->lkeys) {
    GemmArgs g{};
    g.A = h->enc_out; g.lda = RV_E; g.Bm = h->W_mem; g.ldb = RV_U; g.C = h->keys; g.ldc = RV_U;
    g.M = B * Tm; g.N = RV_U; g.K = RV_E; g.row_mask = h->mask;
    Scope sc(h, "gemm_keys");
    launch_gemm_f32(g, false, s);
  }

  // ---- decode loop
  DecState d = h->dec_st;
  d.B = B; d.W = greedy ? 1 : W; d.Tm = Tm; d.V = V; d.L = L;
  d.greedy = greedy; d.attention = c.attention;
  d.start_token = c.start_token; d.end_token = c.end_token; d.pad_token = c.pad_token;
  d.attend_threads = h->opt_att_nt;
  d.keys = h->keys; d.values = h->enc_out; d.mask = h->mask;
  d.W_att = h->W_att; d.W_fc = h->W_fc; d.b_fc = h->b_fc; d.W_q = h->W_q; d.v_att = h->v_att;
  d.call_bases = nullptr; d.call_probs = nullptr; d.call_len = nullptr;
  if (calls) {
    d.call_bases = h->d_bases; d.call_probs = h->d_probs; d.call_len = h->d_clen;
    for (int v = 0; v < RV_MAX_VOCAB; ++v) d.lut[v] = v < V ? lut[v] : 0;
  }
  const int N = B * d.W;
  if (h->opt_taps) {
    const size_t need = (size_t)steps * N * Tm;
    if (need > h->step_align_cap) {
      if (h->step_align) hipFree(h->step_align);
      h->step_align = nullptr; h->step_align_cap = 0;
      for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);   // captured args hold the old pointer
      h->graphs.clear();
      HIPCHK(h, hipMalloc((void**)&h->step_align, need * sizeof(float)));
      h->step_align_cap = need;
    }
    d.step_align = h->step_align;
  } else {
    d.step_align = nullptr;
    if (!greedy && !h->opt_ptaps) d.step_logits = nullptr;
  }

  // The slab decodes as `nsplit` independent sub-slabs on concurrent streams (inside one hipGraph):
  // while one sub-slab is in its HBM-bound attention sweep another runs its latency-bound cell /
  // output / beam phases.  Chunks never interact, and a sub-slab that finishes early is extended by
  // k_dec_finalize exactly as the reference's whole-slab loop would (beam search only; greedy rows
  // keep sampling after their end token, so greedy decodes as one piece).
  int nsplit = (greedy || h->opt_taps || B < 64) ? 1 : std::min(std::max(h->opt_split, 1), 4);
  d.chunk_steps = nullptr;
  // (sizes the decode's LDS)  Luong: 1 = scores and context on the matrix pipe, 2 = the cell product and the output layer too; Bahdanau: 2 = the
  // context, the processed query, the cell product and the output layer on the matrix pipe (the tanh scores stay on the VALU), else packed FMAs
  // Two cells (Luong): 2 = every product of both cells on the matrix pipe, else packed FMAs.
  d.mx_attention = (h->opt_mx_att && d.depth <= 2) ? (h->opt_mx_cell ? 2 : (c.attention == RV_ATT_LUONG && d.depth == 1 ? 1 : 0)) : 0;
  d.Wc16 = h->d_Wc16; d.mx_cdescale = h->mx_cdescale; d.Wl16 = h->d_Wl16; d.mx_ldescale = h->mx_ldescale;
  d.Wq16 = h->d_Wq16; d.mx_qdescale = h->mx_qdescale; d.W1c16 = h->d_W1c16; d.mx_c1descale = h->mx_c1descale;
  h->lpersist = (persist_ok && dec_persist_supported(d)) ? 1 : 0;
  if (persist_ok && !h->lpersist) return fail(h, RV_ESTATE, "internal: persistent decode predicate mismatch");
  if (h->lpersist) { nsplit = 1; d.chunk_steps = h->d_chunk_steps; }
  h->lsplit = nsplit;
  if (!h->lpersist) {
    for (int k = 0; k < d.depth; ++k) {     // get_initial_state: all zeros (basecaller.py:305); the persistent kernel keeps state in LDS
      HIPCHK(h, hipMemsetAsync(d.xh + k * d.ls_xh, 0, sizeof(float) * N * RV_E, s));
      HIPCHK(h, hipMemsetAsync(d.c + k * d.ls_c, 0, sizeof(float) * N * RV_U, s));
    }
  }
  const int Wd = d.W;
  DecState part[4];
  DecParts parts{};
  parts.n = nsplit; parts.steps = steps; parts.S_dev = d.S_dev; parts.S_host = d.S_host;
  for (int g = 0; g < nsplit; ++g) {
    const size_t b0 = (size_t)B * g / nsplit, b1 = (size_t)B * (g + 1) / nsplit;
    DecState& p = part[g];
    p = d;
    p.part = g; p.B = (int)(b1 - b0);
    if (p.call_bases) { p.call_bases += b0 * steps; p.call_probs += b0 * steps; p.call_len += b0; }
    p.keys += b0 * Tm * RV_U; p.values += b0 * Tm * RV_E; p.mask += b0 * Tm;
    p.xh += b0 * Wd * RV_E; p.z += b0 * Wd * RV_G;
    p.c += b0 * Wd * RV_U; p.c_new += b0 * Wd * RV_U; p.h_new += b0 * Wd * RV_U;
    p.tok += b0 * Wd; p.log_probs += b0 * Wd; p.finished += b0 * Wd; p.lengths += b0 * Wd;
    p.step_ids += (size_t)steps * Wd * b0; p.parent_ids += (size_t)steps * Wd * b0; p.step_scores += (size_t)steps * Wd * b0;
    if (p.step_logits) p.step_logits += (size_t)steps * Wd * V * b0;
    if (p.step_align) p.step_align += (size_t)steps * Wd * Tm * b0;
    p.nfin = d.nfin + g * (c.max_output_len + 1);
    parts.nfin[g] = p.nfin; parts.B[g] = p.B;
    if (!h->lpersist) launch_dec_init(p, s);
  }
  auto enqueue_steps = [&](bool profiled) -> int {
    for (int g = 1; g < nsplit; ++g) {
      HIPCHK(h, hipEventRecord(h->ev_fork, s));
      HIPCHK(h, hipStreamWaitEvent(side_stream(h, g - 1), h->ev_fork, 0));
    }
    for (int g = 0; g < nsplit; ++g) launch_decode_steps(h, part[g], g == 0 ? s : side_stream(h, g - 1), profiled && nsplit == 1);
    for (int g = 1; g < nsplit; ++g) {
      HIPCHK(h, hipEventRecord(h->ev_join[g - 1], side_stream(h, g - 1)));
      HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[g - 1], 0));
    }
    return RV_OK;
  };
  if (h->lpersist) {
    part[0] = d; part[0].part = 0; parts.n = 1;
    if (greedy) HIPCHK(h, hipMemsetAsync(d.nfin, 0, 2 * sizeof(int), s));   // chunks-finished count and latest first-finish step
    {   // attention memory in the form the persistent decode keeps on chip: [keys | U'] = enc_out . [W_mem | A_c]
      GemmArgs g{};
      g.A = h->enc_out; g.lda = RV_E; g.Bm = h->d_Wmp; g.ldb = RV_E; g.C = h->mem2; g.ldc = RV_E;
      g.M = B * Tm; g.N = RV_E; g.K = RV_E;
      g.xcd_remap = 1;
      Scope sc(h, "gemm_memory");
      if (h->opt_split_proj) launch_gemm_mem_split(h->enc_out, B * Tm, h->d_Wmp16, h->mem2, s);
      else launch_gemm_f32(g, false, s);
    }
    d.values = h->mem2; part[0].values = h->mem2;
    d.mx_kscale = h->mx_kscale; d.mx_kdescale = std::ldexp(1.0f, -14) / h->mx_kscale;
    d.mx_uscale = h->mx_uscale; d.mx_udescale = std::ldexp(1.0f, -14) / h->mx_uscale;
    Scope sc(h, "dec_persist", nullptr, true);
    launch_dec_persist(d, d.depth > 1 ? h->dec[0].W + (size_t)V * RV_G : h->d_Wcat2, h->dec[0].W, h->dec[0].b,
                       d.depth > 1 ? h->dec[1].W : nullptr, d.depth > 1 ? h->dec[1].b : nullptr, h->d_Nh, s);
  } else if (h->opt_graph && h->opt_profile != 2) {
    GraphKey key{B, d.W, Tm, L, greedy ? 1 : 0, h->opt_taps * 2 + h->lflash + 4 * h->opt_att_nt + 4096 * (d.step_logits ? 1 : 0), nsplit};   // every captured pointer that can change is in the key
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
      if (h->graphs.size() >= 32) {      // bound the cache (callers with ever-changing slab shapes)
        for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
        h->graphs.clear();
      }
      hipGraph_t graph = nullptr;
      HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      const int rc = enqueue_steps(false);
      hipError_t ce = hipStreamEndCapture(s, &graph);
      if (rc != RV_OK) return rc;
      HIPCHK(h, ce);
      hipGraphExec_t exec = nullptr;
      HIPCHK(h, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      hipGraphDestroy(graph);
      it = h->graphs.emplace(key, exec).first;
    }
    Scope sc(h, "decode_graph", nullptr, true);
    HIPCHK(h, hipGraphLaunch(it->second, s));
  } else {
    if (h->opt_profile == 2) nsplit = 1, parts.n = 1;     // per-kernel events need one stream
    if (nsplit == 1) { part[0] = d; part[0].part = 0; part[0].nfin = d.nfin; parts.nfin[0] = d.nfin; parts.B[0] = B; launch_dec_init(part[0], s); }
    const int rc = enqueue_steps(h->opt_profile == 2);
    if (rc != RV_OK) return rc;
  }

  {
    Scope sc(h, "dec_finalize");
    if (!h->lpersist) launch_dec_reduce_steps(parts, s);
    for (int g = 0; g < parts.n; ++g) {
      const size_t b0 = parts.n == 1 ? 0 : (size_t)B * g / parts.n;
      launch_dec_finalize(part[g], tk + b0 * steps, o2 + b0 * steps * (greedy ? V : 1), s, ptab);     // (ptab: persistent decode only, one part)
    }
  }
  // (S reaches the host through d.S_host: the finalize / reduce kernel stores it straight into the mapped pinned word)
  if (!dev_out && !calls) {
    HIPCHK(h, hipMemcpyAsync(h->pin_tok, tk, sizeof(int32_t) * B * steps, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_out2, o2, sizeof(float) * B * steps * (greedy ? V : 1), hipMemcpyDeviceToHost, s));
  }
  if (calls) {
    HIPCHK(h, hipMemcpyAsync(h->pin_bases, h->d_bases, (size_t)B * steps, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_probs, h->d_probs, sizeof(float) * B * steps, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_clen, h->d_clen, sizeof(int) * B, hipMemcpyDeviceToHost, s));
  }
  h->lW = d.W;
  return RV_OK;
}

// The rest of a call: wait for the context's stream, hand the staged results to the caller's host buffers.
int finish(RvContext* h, int32_t* tokens, float* out2, const CallsOut* calls, int32_t* S_out) {
  if (!h || !S_out) return RV_EINVAL;
  if (!h->pend.busy) return fail(h, RV_ESTATE, "no call to collect on this context");
  const RvContext::PendingCall p = h->pend;
  h->pend.busy = false;
  *S_out = 0;
  if (p.trivial) return RV_OK;
  if (!p.dev_out && !p.calls && (!tokens || !out2)) return fail(h, RV_EINVAL, "null output pointer");
  if (p.calls && (!calls || !calls->bases || !calls->lengths || !calls->probs)) return fail(h, RV_EINVAL, "null calls output pointer");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  const int S = *h->pin_S;
  const size_t B = p.B, steps = p.steps;
  if (!p.dev_out && !p.calls) {
    memcpy(tokens, h->pin_tok, sizeof(int32_t) * B * steps);
    memcpy(out2, h->pin_out2, sizeof(float) * B * steps * (p.greedy ? p.V : 1));
  }
  if (p.calls) {
    memcpy(calls->bases, h->pin_bases, B * steps);
    memcpy(calls->probs, h->pin_probs, sizeof(float) * B * steps);
    memcpy(calls->lengths, h->pin_clen, sizeof(int) * B);
  }
  drain_profile(h);
  *S_out = S;
  h->lS = S;
  return RV_OK;
}

void sync_child(RvContext* k, const RvContext* p);

int run(RvContext* h, const float* raw, const float* ev, bool dev_in, int B, int T_r, int T_e, int W,
        int L, bool greedy, int32_t* tokens, float* out2, bool dev_out, int32_t* S_out, const CallsOut* calls = nullptr) {
  if (!h) return RV_EINVAL;
  if (!S_out || (B > 0 && L > 1 && !calls && (!tokens || !out2))) return fail(h, RV_EINVAL, "null output pointer");
  if (calls && (!calls->lut || !calls->bases || !calls->lengths || !calls->probs)) return fail(h, RV_EINVAL, "null calls output pointer");
  *S_out = 0;
  // A synchronous call between asynchronous ones: it runs on the handle's own context when that one is idle, else on any idle
  // context of the handle -- never on one that still holds an uncollected ticket (its stream work and output pointers are live).
  // Debug taps and rv_get_tensor belong to the handle's own context, so with taps on that context must be the idle one.
  RvContext* ctx = h;
  if (h->pend.busy) {
    if (h->opt_taps || h->opt_ptaps)
      return fail(h, RV_ESTATE, "debug taps need the handle's own context, which holds an uncollected asynchronous call: collect ticket %d first", h->pend.ticket);
    ctx = nullptr;
    for (RvContext* k : h->kids) if (!k->pend.busy) { ctx = k; break; }
    if (!ctx) return fail(h, RV_ESTATE, "every slab context of this handle holds an uncollected asynchronous call: collect one before a synchronous call");
  }
  h->inflight_hint = 1;
  if (ctx != h) sync_child(ctx, h);
  const int rc = enqueue(ctx, raw, ev, dev_in, B, T_r, T_e, W, L, greedy, tokens, out2, dev_out, calls ? calls->lut : nullptr);
  if (rc != RV_OK) { ctx->pend.busy = false; if (ctx != h) h->err = ctx->err; return rc; }   // (ctx was idle on entry: the flag is this call's)
  const int rf = finish(ctx, tokens, out2, calls, S_out);
  if (ctx != h) { if (rf != RV_OK) h->err = ctx->err; h->lS = ctx->lS; }
  return rf;
}


// Per-slab working set of a context: inputs, activations, decoder state, outputs, staging, side streams.  Weights are not part of it
// (child contexts of the asynchronous calls share their parent's).
int alloc_slab_buffers(RvContext* h) {
  const RvConfig& c = h->cfg;
#define TRY(x) do { int r_ = (x); if (r_ != RV_OK) return r_; } while (0)
#define HIPTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(h, RV_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
  const bool use_raw = c.mode != RV_MODE_EVENT, use_ev = c.mode != RV_MODE_RAW;
  const size_t B = c.max_batch, Tr = use_raw ? c.max_raw_len : 0, Te = use_ev ? c.max_event_len : 0;
  const size_t Tm = Tr + Te, L = c.max_output_len, N = B * c.max_beam, V = c.vocab;
  TRY(dalloc(h, &h->d_raw, B * Tr));
  TRY(dalloc(h, &h->d_ev, B * Te * 5));
  TRY(dalloc(h, &h->mask, B * Tm));
  const int npp = c.enc_depth > 2 ? 2 : (c.enc_depth > 1 ? 1 : 0);
  for (int p = 0; p < npp; ++p) {
    if (use_raw) TRY(dalloc(h, &h->act[0][p], B * Tr * RV_E));
    if (use_ev) TRY(dalloc(h, &h->act[1][p], B * Te * RV_E));
  }
  // The pre-projected tensors xw [max_batch, T_max, 2, 512] (4 KB per chunk-timestep) of the matrix-pipe recurrence (the default) and of the
  // unfused packed-FMA path: allocated HERE, with the context -- not on a context's first call.  (Round 4 found the first-use hipMalloc of a
  // fresh context inside bench.py's timed region: the driver's 5 warm-up steps touch 5 of the 10 contexts, the other five allocated 2 x ~300 MB
  // each while the GPU drained -- 1.5 to 10 ms of host stall per context, which rounds 3 and 4 had read as the pipeline's fill time.)
  if (use_raw && c.enc_depth > 1) TRY(dalloc(h, &h->xw[0], B * Tr * 2 * RV_G));
  if (use_ev) TRY(dalloc(h, &h->xw[1], B * Te * 2 * RV_G));
  for (int e = 0; e < 2; ++e)
    for (int st = 0; st < 2; ++st)
      for (int k = 0; k < 4; ++k) TRY(dalloc(h, &h->st[e][st][k], B * RV_U));
  TRY(dalloc(h, &h->enc_out, B * Tm * RV_E));
  TRY(dalloc(h, &h->keys, B * Tm * RV_U));
  TRY(dalloc(h, &h->mem2, B * Tm * RV_E));
  DecState& d = h->dec_st;
#ifdef RV_DIAG   // timing ablations that INVALIDATE results: only in `make diag` builds (libravvent_hip_diag.so), never in the product library
  if (const char* e = getenv("RV_ATT_STOP")) d.dbg_stop = atoi(e);
#endif
  if (getenv("RV_REC_STAMPS")) { TRY(dalloc(h, &h->rec_ts, 24)); h->rec_ts_layer = atoi(getenv("RV_REC_STAMPS")) == 2 ? 1 : 0; }
  if (getenv("RV_DBG_STAMPS")) TRY(dalloc(h, &d.dbg_ts, 16));   // diagnostic builds: in-kernel phase stamps   // timing ablation only; results are invalid when set
  d.depth = c.dec_depth; d.ls_xh = N * RV_E; d.ls_c = N * RV_U;
  TRY(dalloc(h, &d.xh, c.dec_depth * N * RV_E));
  TRY(dalloc(h, &d.z, N * RV_G));
  TRY(dalloc(h, &d.c, c.dec_depth * N * RV_U));
  TRY(dalloc(h, &d.c_new, c.dec_depth * N * RV_U));
  TRY(dalloc(h, &d.h_new, c.dec_depth * N * RV_U));
  TRY(dalloc(h, &d.tok, N));
  TRY(dalloc(h, &d.log_probs, N));
  TRY(dalloc(h, &d.finished, N));
  TRY(dalloc(h, &d.lengths, N));
  TRY(dalloc(h, &d.step_ids, L * N));
  TRY(dalloc(h, &d.parent_ids, L * N));
  TRY(dalloc(h, &d.step_scores, L * N));
  TRY(dalloc(h, &d.step_logits, L * N * V));
  TRY(dalloc(h, &d.nfin, 4 * (L + 1)));
  TRY(dalloc(h, &d.S_dev, 8));
  TRY(dalloc(h, &h->d_chunk_steps, (size_t)c.max_batch));
  // (side streams are created on first use -- side_stream(): the contexts of the asynchronous calls never need them, and every
  //  stream of the process competes for the same few hardware queues)
  for (int g = 0; g < 3; ++g) HIPTRY(hipEventCreateWithFlags(&h->ev_join[g], hipEventDisableTiming));
  HIPTRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  if (const char* e = getenv("RV_DECODE_SPLIT")) h->opt_split = atoi(e);
  TRY(dalloc(h, &h->out_tokens, B * L));
  TRY(dalloc(h, &h->out2, B * L * V));
  HIPTRY(hipHostMalloc((void**)&h->pin_raw, std::max<size_t>(B * Tr, 1) * sizeof(float), hipHostMallocDefault));
  HIPTRY(hipHostMalloc((void**)&h->pin_ev, std::max<size_t>(B * Te * 5, 1) * sizeof(float), hipHostMallocDefault));
  HIPTRY(hipHostMalloc((void**)&h->pin_tok, B * L * sizeof(int32_t), hipHostMallocDefault));
  HIPTRY(hipHostMalloc((void**)&h->pin_out2, B * L * V * sizeof(float), hipHostMallocDefault));
  HIPTRY(hipHostMalloc((void**)&h->pin_S, sizeof(int), hipHostMallocMapped));
  HIPTRY(hipHostGetDevicePointer((void**)&d.S_host, h->pin_S, 0));
  TRY(dalloc(h, &h->d_bases, B * L));
  TRY(dalloc(h, &h->d_probs, B * L));
  TRY(dalloc(h, &h->d_clen, B));
  HIPTRY(hipHostMalloc((void**)&h->pin_bases, B * L, hipHostMallocDefault));
  HIPTRY(hipHostMalloc((void**)&h->pin_probs, B * L * sizeof(float), hipHostMallocDefault));
  HIPTRY(hipHostMalloc((void**)&h->pin_clen, B * sizeof(int), hipHostMallocDefault));
  // the table lives in mapped pinned memory: the three kernels that need a caller address read it over the host link once per workgroup
  // (a device copy refreshed by a memcpy node in front of the graph measured the same, and is one node more)
  HIPTRY(hipHostMalloc((void**)&h->pin_ptab, sizeof(void*) * RV_PTAB_N, hipHostMallocMapped));
  HIPTRY(hipHostGetDevicePointer((void**)&h->d_ptab, h->pin_ptab, 0));
#undef TRY
#undef HIPTRY
  return RV_OK;
}

// A further slab context of handle p for the asynchronous calls: own stream and buffers, p's weights.
int create_child(RvContext* p, RvContext** out) {
  RvContext* h = new RvContext();
  h->cfg = p->cfg;
  h->parent = p;
  // (stream priorities -- the contexts at the runtime's three levels in turn, so that slabs submitted together leave lockstep -- measured
  //  no gain at the driver's 20 steps and 4 % less once the stream has settled: tools/stream_ab.py, round 4)
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(p, RV_EHIP, "hipStreamCreate failed for an asynchronous context"); }
  h->d_w = p->d_w; h->n_w = p->n_w;
  h->d_WmemT = p->d_WmemT; h->d_Up = p->d_Up; h->d_Wp = p->d_Wp; h->d_Wsb = p->d_Wsb; h->d_Wh = p->d_Wh; h->d_Ua = p->d_Ua;
  h->d_Wx16 = p->d_Wx16; h->d_bx2 = p->d_bx2; h->d_Wmp = p->d_Wmp; h->d_Wcat2 = p->d_Wcat2; h->d_Nh = p->d_Nh; h->d_Wmp16 = p->d_Wmp16; h->d_Wc16 = p->d_Wc16; h->d_Wl16 = p->d_Wl16;
  h->d_Wq16 = p->d_Wq16; h->d_W1c16 = p->d_W1c16;
  h->d_WcatT = p->d_WcatT;
  bind_weights(h);
  const int rc = alloc_slab_buffers(h);
  if (rc != RV_OK) { p->err = h->err; rv_destroy(h); return rc; }
  *out = h;
  return RV_OK;
}

// options and weight-derived scalars of the parent, as of now
void sync_child(RvContext* k, const RvContext* p) {
  k->loaded = p->loaded; k->mx_kscale = p->mx_kscale; k->mx_uscale = p->mx_uscale;
  k->opt_split = p->opt_split; k->opt_att_nt = p->opt_att_nt; k->opt_side_ev = p->opt_side_ev; k->opt_lane_inproj = p->opt_lane_inproj; k->opt_persist = p->opt_persist;
  k->opt_flash = p->opt_flash; k->opt_split_proj = p->opt_split_proj; k->opt_mx_att = p->opt_mx_att; k->opt_mx_cell = p->opt_mx_cell; k->mx_cdescale = p->mx_cdescale; k->mx_ldescale = p->mx_ldescale; k->opt_tail_wave = p->opt_tail_wave;
  k->mx_qdescale = p->mx_qdescale; k->mx_c1descale = p->mx_c1descale;
  k->opt_fuse = p->opt_fuse; k->opt_wide = p->opt_wide; k->opt_graph = p->opt_graph; k->opt_profile = p->opt_profile;
  k->opt_taps = 0; k->opt_ptaps = 0;      // debug taps belong to the synchronous calls
  k->inflight_hint = p->inflight_hint;
}

}  // namespace

// Loading the library is the earliest point a C caller reaches: ask the HIP runtime for 16 hardware queues (the asynchronous calls keep up
// to 16 slab contexts = streams in flight) unless the environment already says something, or RAVVENT_KEEP_ENV opts out.  It takes effect
// when the runtime starts (the process's first HIP call); rv_set_option("async_depth") reports when the setting in force is too small.
__attribute__((constructor)) static void rv_default_hw_queues() {
  if (!getenv("RAVVENT_KEEP_ENV")) setenv("GPU_MAX_HW_QUEUES", "16", 0);
}

extern "C" {

int rv_abi_version(void) { return RV_ABI_VERSION; }

const char* rv_last_error(rv_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rv_create(const RvConfig* cfg, rv_handle* out) {
  if (!cfg || !out) return fail(nullptr, RV_EINVAL, "null argument");
  *out = nullptr;
  const RvConfig& c = *cfg;
  if (c.enc_units != RV_U || c.dec_units != RV_U)
    return fail(nullptr, RV_EUNSUPPORTED, "this build is specialised for enc_units = dec_units = 128 (got %d, %d)", c.enc_units, c.dec_units);
  if (c.dec_depth < 1 || c.dec_depth > 4)
    return fail(nullptr, RV_EUNSUPPORTED, "decoder_depth %d outside [1,4]", c.dec_depth);
  if (c.enc_depth < 1 || c.enc_depth > 8) return fail(nullptr, RV_EINVAL, "encoder_depth %d outside [1,8]", c.enc_depth);
  if (c.vocab < 2 || c.vocab > RV_MAX_VOCAB) return fail(nullptr, RV_EINVAL, "vocab %d outside [2,%d]", c.vocab, RV_MAX_VOCAB);
  if (c.mode < 0 || c.mode > 2 || c.attention < 0 || c.attention > 1) return fail(nullptr, RV_EINVAL, "bad mode/attention");
  if (c.max_beam < 1 || c.max_beam > RV_MAX_BEAM) return fail(nullptr, RV_EINVAL, "max_beam %d outside [1,%d]", c.max_beam, RV_MAX_BEAM);
  if (c.max_batch < 1 || c.max_raw_len < 1 || c.max_event_len < 1 || c.max_output_len < 1)
    return fail(nullptr, RV_EINVAL, "maximum shapes must be positive");
  for (int t : {c.start_token, c.end_token, c.pad_token})
    if (t < 0 || t >= c.vocab) return fail(nullptr, RV_EINVAL, "token id %d outside vocab", t);

  {   // layer 0 keeps a workgroup's input windows in LDS (160 KB): reject maximum shapes no kernel variant can launch
    const int bt = pick_rows_per_block(c.max_batch);
    if (c.mode != RV_MODE_EVENT && !lstm_rec_window_fits(1, bt, c.max_raw_len))
      return fail(nullptr, RV_EUNSUPPORTED, "max_raw_len %d with max_batch %d (%d chunks per workgroup) exceeds the 160 KB of LDS the layer-0 recurrence stages its input windows in", c.max_raw_len, c.max_batch, bt);
    if (c.mode != RV_MODE_RAW && !lstm_rec_window_fits(5, bt, c.max_event_len))
      return fail(nullptr, RV_EUNSUPPORTED, "max_event_len %d with max_batch %d (%d chunks per workgroup) exceeds the 160 KB of LDS the layer-0 recurrence stages its input windows in", c.max_event_len, c.max_batch, bt);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, RV_EHIP, "no HIP device visible: libravvent_hip has no CPU path");
  if (c.device < 0 || c.device >= ndev) return fail(nullptr, RV_EINVAL, "device %d outside [0,%d)", c.device, ndev);

  RvContext* h = new RvContext();
  h->cfg = c;
  auto bail = [&](int code) { g_create_error = h->err; rv_destroy(h); return code; };
#define TRY(x) do { int r_ = (x); if (r_ != RV_OK) return bail(r_); } while (0)
#define HIPTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail(h, RV_EHIP, "%s: %s", #x, hipGetErrorString(e_)); return bail(RV_EHIP); } } while (0)
  HIPTRY(hipSetDevice(c.device));
  HIPTRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPTRY(configure_decode_kernels());    // a refused LDS opt-in would otherwise surface as an opaque launch error later
  HIPTRY(configure_rec_kernels());
  HIPTRY(configure_mx_kernels());
  HIPTRY(configure_gemm_kernels());
  h->n_w = weight_count(c);
  TRY(dalloc(h, &h->d_w, h->n_w));
  TRY(dalloc(h, &h->d_WcatT, (size_t)c.dec_depth * RV_G * RV_E));
  TRY(dalloc(h, &h->d_Wmp, (size_t)RV_E * RV_E));
  TRY(dalloc(h, &h->d_Wmp16, RV_WMP16_SLOT));
  TRY(dalloc(h, &h->d_Wc16, (size_t)2 * 3 * RV_U * RV_G));      // (two cells: 384 rows)
  if (c.dec_depth == 2) TRY(dalloc(h, &h->d_W1c16, (size_t)2 * RV_E * RV_G));
  TRY(dalloc(h, &h->d_Wl16, (size_t)2 * RV_E * 16));
  TRY(dalloc(h, &h->d_Wq16, (size_t)2 * RV_U * RV_U));
  TRY(dalloc(h, &h->d_Wcat2, (size_t)RV_E * RV_G));
  TRY(dalloc(h, &h->d_Nh, (size_t)RV_U * RV_MAX_VOCAB));
  TRY(dalloc(h, &h->d_Up, (size_t)2 * c.enc_depth * 2 * RV_U * RV_G));
  TRY(dalloc(h, &h->d_Ua, (size_t)2 * c.enc_depth * 2 * RV_UA_SLOT));
  if (c.enc_depth > 1) TRY(dalloc(h, &h->d_Wx16, (size_t)2 * (c.enc_depth - 1) * RV_WX16_SLOT));
  if (c.enc_depth > 1) TRY(dalloc(h, &h->d_bx2, (size_t)2 * (c.enc_depth - 1) * 2 * RV_G));
  if (c.enc_depth > 1) TRY(dalloc(h, &h->d_Wp, (size_t)2 * (c.enc_depth - 1) * 2 * RV_E * RV_G));
  if (c.enc_depth > 1) TRY(dalloc(h, &h->d_Wh, (size_t)2 * (c.enc_depth - 1) * 2 * RV_WH_SLOT));
  if (c.enc_depth > 1) TRY(dalloc(h, &h->d_Wsb, (size_t)2 * (c.enc_depth - 1) * 2 * 3 * RV_E * RV_G));
#ifdef RV_DIAG
  if (const char* e = getenv("RV_DBG_ROLE")) h->dbg_role = atoi(e);
#endif
  TRY(dalloc(h, &h->d_WmemT, (size_t)RV_U * RV_E));
  bind_weights(h);
  TRY(alloc_slab_buffers(h));
#undef TRY
#undef HIPTRY
  *out = h;
  return RV_OK;
}

void rv_destroy(rv_handle h) {
  if (!h) return;
  for (RvContext* k : h->kids) rv_destroy(k);
  h->kids.clear();
  hipSetDevice(h->cfg.device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
  for (auto& kv : h->slab_graphs) hipGraphExecDestroy(kv.second.exec);
  for (auto& p : h->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  for (auto e : h->ev_pool) hipEventDestroy(e);
  for (int g = 0; g < 3; ++g) { if (h->side[g]) hipStreamDestroy(h->side[g]); if (h->ev_join[g]) hipEventDestroy(h->ev_join[g]); }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->step_align) hipFree(h->step_align);
  for (void* p : {(void*)h->pin_raw, (void*)h->pin_ev, (void*)h->pin_tok, (void*)h->pin_out2, (void*)h->pin_S, (void*)h->pin_bases, (void*)h->pin_probs, (void*)h->pin_clen, (void*)h->pin_ptab}) if (p) hipHostFree(p);
  for (void* p : h->allocs) hipFree(p);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

size_t rv_weight_count(rv_handle h) { return h ? h->n_w : 0; }

int rv_load_weights(rv_handle h, const float* blob, size_t n_floats) {
  if (!h) return RV_EINVAL;
  if (!blob) return fail(h, RV_EINVAL, "null weight blob");
  if (n_floats != h->n_w) return fail(h, RV_EINVAL, "weight blob has %zu floats, config needs %zu", n_floats, h->n_w);
  if (h->pend.busy) return fail(h, RV_ESTATE, "collect the calls in flight before loading weights");
  for (RvContext* k : h->kids) if (k->pend.busy) return fail(h, RV_ESTATE, "collect the calls in flight before loading weights");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  h->opt_gen++;                      // (weight-derived scalars travel by value in the decode's kernel arguments)
  HIPCHK(h, hipMemcpyAsync(h->d_w, blob, n_floats * sizeof(float), hipMemcpyHostToDevice, h->stream));
  {   // derived layout for the decoder cell kernel: the input-kernel rows that multiply a dense input
      // (rows V.. of W_0; all of W_k for k >= 1) followed by U_k form a contiguous [256][512] matrix in
      // the blob; the kernel wants it column-major ([512][256]).
    std::vector<float> t((size_t)RV_G * RV_E);
    for (int l = 0; l < h->cfg.dec_depth; ++l) {
      const size_t off = (size_t)(h->dec[l].W - h->d_w) + (l == 0 ? (size_t)h->cfg.vocab * RV_G : 0);
      for (int k = 0; k < RV_E; ++k)
        for (int n = 0; n < RV_G; ++n) t[(size_t)n * RV_E + k] = blob[off + (size_t)k * RV_G + n];
      HIPCHK(h, hipMemcpy(h->d_WcatT + (size_t)l * RV_G * RV_E, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // every encoder layer: U [128][512] -> [32 i][512 threads][4 slots] (thread = 4 j + kq holds k = 32 kq + i, slot r = gate (kq + r) & 3)
    for (int e = 0; e < 2; ++e)
      for (int l = 0; l < h->cfg.enc_depth; ++l)
        for (int dr = 0; dr < 2; ++dr) {
          const size_t off = (size_t)(h->enc[e][l][dr].U - h->d_w);
          std::vector<float> up((size_t)RV_U * RV_G);
          for (int i = 0; i < 32; ++i)
            for (int tid = 0; tid < 512; ++tid) {
              const int j = tid >> 2, kq = tid & 3;
              for (int r = 0; r < 4; ++r)
                up[((size_t)i * 512 + tid) * 4 + r] = blob[off + (size_t)(32 * kq + i) * RV_G + ((kq + r) & 3) * RV_U + j];
            }
          float* dst = h->d_Up + ((size_t)(e * h->cfg.enc_depth + l) * 2 + dr) * RV_U * RV_G;
          HIPCHK(h, hipMemcpy(dst, up.data(), up.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    // ... and as MFMA A fragments of U^T for the matrix-pipe recurrence: wave w, gate g, k-step ks, part p, lane (m = lane%16, kq = lane/16)
    // holds s_r U[32 ks + 8 kq + j][r], r = 128 g + 16 w + m (s_r: power of two, the column's largest element into [2^13, 2^14))
    for (int e = 0; e < 2; ++e)
      for (int l = 0; l < h->cfg.enc_depth; ++l)
        for (int dr = 0; dr < 2; ++dr) {
          const size_t off = (size_t)(h->enc[e][l][dr].U - h->d_w);
          std::vector<uint16_t> ua(RV_UA_SLOT);
          float cs[RV_G];
          for (int r = 0; r < RV_G; ++r) {
            float mx = 0.f;
            for (int k = 0; k < RV_U; ++k) mx = std::max(mx, std::fabs(blob[off + (size_t)k * RV_G + r]));
            int ex = 0;
            if (mx > 0.f && std::isfinite(mx)) std::frexp(mx, &ex);
            cs[r] = std::ldexp(1.0f, 14 - ex);
          }
          for (int w = 0; w < 8; ++w)
            for (int g = 0; g < 4; ++g)
              for (int ks = 0; ks < 4; ++ks)
                for (int ln = 0; ln < 64; ++ln)
                  for (int j = 0; j < 8; ++j) {
                    const int r = g * RV_U + 16 * w + (ln & 15);
                    const float v = blob[off + (size_t)(32 * ks + 8 * (ln >> 4) + j) * RV_G + r] * cs[r];   // exact
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
                    const size_t base = ((((size_t)w * 4 + g) * 4 + ks) * 2) * 64 * 8;
                    ua[base + (size_t)ln * 8 + j] = hb;
                    ua[base + 64 * 8 + (size_t)ln * 8 + j] = lb;
                  }
          for (int r = 0; r < RV_G; ++r) { const float f = std::ldexp(1.0f, -14) / cs[r]; memcpy(&ua[(size_t)2 * RV_U * RV_G + 2 * r], &f, 4); }
          HIPCHK(h, hipMemcpy(h->d_Ua + ((size_t)(e * h->cfg.enc_depth + l) * 2 + dr) * RV_UA_SLOT, ua.data(), ua.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
    // encoder layers >= 1, both directions' input kernels as the split GEMM's B slabs: column block cb = 2 dir + half holds columns
    // 256 half .. of direction dir; [cb][8 k-steps][16 tiles][2 parts][64 lanes][8 f16] of the column-scaled kernel, then 1024 factors
    for (int e = 0; e < 2; ++e)
      for (int l = 1; l < h->cfg.enc_depth; ++l) {
        std::vector<uint16_t> img(RV_WX16_SLOT);
        std::vector<float> b2(2 * RV_G);
        for (int dr = 0; dr < 2; ++dr) {
          const size_t off = (size_t)(h->enc[e][l][dr].W - h->d_w), boff = (size_t)(h->enc[e][l][dr].b - h->d_w);
          for (int n = 0; n < RV_G; ++n) b2[(size_t)dr * RV_G + n] = blob[boff + n];
          for (int n = 0; n < RV_G; ++n) {
            float mx = 0.f;
            for (int k = 0; k < RV_E; ++k) mx = std::max(mx, std::fabs(blob[off + (size_t)k * RV_G + n]));
            int ex = 0;
            if (mx > 0.f && std::isfinite(mx)) std::frexp(mx, &ex);
            const float sc = std::ldexp(1.0f, 14 - ex);
            const int cb = 2 * dr + n / RV_E, nn = n % RV_E;
            const float f = std::ldexp(1.0f, -14) / sc;
            memcpy(&img[(size_t)4 * 2 * RV_E * RV_E + 2 * ((size_t)cb * RV_E + nn)], &f, 4);
            for (int k = 0; k < RV_E; ++k) {
              const float v = blob[off + (size_t)k * RV_G + n] * sc;
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (float)hi);
              uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
              const int ks = k / 32, ln = 16 * ((k % 32) / 8) + (nn % 16), j = k % 8, nt = nn / 16;
              const size_t base = (size_t)cb * 2 * RV_E * RV_E + ((((size_t)ks * 16 + nt) * 2) * 64) * 8;
              img[base + (size_t)ln * 8 + j] = hb;
              img[base + 64 * 8 + (size_t)ln * 8 + j] = lb;
            }
          }
        }
        const size_t slot = (size_t)(e * (h->cfg.enc_depth - 1) + (l - 1));
        HIPCHK(h, hipMemcpy(h->d_Wx16 + slot * RV_WX16_SLOT, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->d_bx2 + slot * 2 * RV_G, b2.data(), b2.size() * sizeof(float), hipMemcpyHostToDevice));
      }
    // encoder layers >= 1: W [256][512] -> [32 column tiles][16 k-groups][64 lanes][4]: lane (q = lane/16, col = lane%16)
    // of tile nt finds W[16 g + 4 i + q][16 nt + col] for its 4 MFMAs i of k-group g in one float4 (lstm_rec.hip)
    for (int e = 0; e < 2; ++e)
      for (int l = 1; l < h->cfg.enc_depth; ++l)
        for (int dr = 0; dr < 2; ++dr) {
          const size_t off = (size_t)(h->enc[e][l][dr].W - h->d_w);
          for (int nt = 0; nt < 32; ++nt)
            for (int g = 0; g < 16; ++g)
              for (int ln = 0; ln < 64; ++ln)
                for (int i = 0; i < 4; ++i)
                  t[(((size_t)nt * 16 + g) * 64 + ln) * 4 + i] = blob[off + (size_t)(16 * g + 4 * i + (ln >> 4)) * RV_G + 16 * nt + (ln & 15)];
          float* dst = h->d_Wp + ((size_t)(e * (h->cfg.enc_depth - 1) + (l - 1)) * 2 + dr) * RV_E * RV_G;
          HIPCHK(h, hipMemcpy(dst, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
          // ... and as bf16 parts: lane (kq = lane/16, col = lane%16) of tile nt, k-step ks holds W[32 ks + 8 kq + j][16 nt + col], j = 0..7
          std::vector<uint16_t> sb((size_t)3 * RV_E * RV_G);
          for (int nt = 0; nt < 32; ++nt)
            for (int ks = 0; ks < 8; ++ks)
              for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                  float r = blob[off + (size_t)(32 * ks + 8 * (ln >> 4) + j) * RV_G + 16 * nt + (ln & 15)];
                  for (int part = 0; part < 3; ++part) {
                    const uint16_t b = bf16_rne(r);
                    sb[((((size_t)nt * 3 + part) * 8 + ks) * 64 + ln) * 8 + j] = b;
                    r -= bf16_value(b);                    // exact
                  }
                }
          // ... and as two f16 parts of s_n W (s_n: power of two, column maximum into [2^13, 2^14]), then the factors 2^-14 / s_n
          std::vector<uint16_t> wh(RV_WH_SLOT);
          float cs[RV_G];
          for (int n = 0; n < RV_G; ++n) {
            float mx = 0.f;
            for (int k = 0; k < RV_E; ++k) mx = std::max(mx, std::fabs(blob[off + (size_t)k * RV_G + n]));
            int ex = 0;
            if (mx > 0.f && std::isfinite(mx)) std::frexp(mx, &ex);     // mx = m 2^ex, m in [0.5, 1)
            cs[n] = std::ldexp(1.0f, 14 - ex);                           // s_n; mx s_n in [2^13, 2^14)
          }
          for (int nt = 0; nt < 32; ++nt)
            for (int ks = 0; ks < 8; ++ks)
              for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                  const int n = 16 * nt + (ln & 15);
                  const float v = blob[off + (size_t)(32 * ks + 8 * (ln >> 4) + j) * RV_G + n] * cs[n];   // exact
                  const _Float16 hi = (_Float16)v;
                  const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);
                  uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
                  wh[((((size_t)nt * 2 + 0) * 8 + ks) * 64 + ln) * 8 + j] = hb;
                  wh[((((size_t)nt * 2 + 1) * 8 + ks) * 64 + ln) * 8 + j] = lb;
                }
          for (int n = 0; n < RV_G; ++n) { const float f = std::ldexp(1.0f, -14) / cs[n]; memcpy(&wh[(size_t)2 * RV_E * RV_G + 2 * n], &f, 4); }
          HIPCHK(h, hipMemcpy(h->d_Wh + ((size_t)(e * (h->cfg.enc_depth - 1) + (l - 1)) * 2 + dr) * RV_WH_SLOT, wh.data(), wh.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
          uint16_t* dsb = h->d_Wsb + ((size_t)(e * (h->cfg.enc_depth - 1) + (l - 1)) * 2 + dr) * 3 * RV_E * RV_G;
          HIPCHK(h, hipMemcpy(dsb, sb.data(), sb.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
    {   // [W_mem | A_c] [256][256]: column block 0 = W_mem (keys), column block 1 = rows 128..383 of the attention layer
        // (its context part), so that enc_out . Wmp = [keys | image of the values under the attention layer]
      const size_t mo = (size_t)(h->W_mem - h->d_w), ao = (size_t)(h->W_att - h->d_w) + (size_t)RV_U * RV_U;
      std::vector<float> wmp((size_t)RV_E * RV_E);
      for (int kk = 0; kk < RV_E; ++kk)
        for (int n = 0; n < RV_U; ++n) {
          wmp[(size_t)kk * RV_E + n] = blob[mo + (size_t)kk * RV_U + n];
          wmp[(size_t)kk * RV_E + RV_U + n] = blob[ao + (size_t)kk * RV_U + n];
        }
      HIPCHK(h, hipMemcpy(h->d_Wmp, wmp.data(), wmp.size() * sizeof(float), hipMemcpyHostToDevice));
      {   // |enc_out| <= 1, so |key_tu| <= sum_k |W_mem[k][u]| and |U'_tn| <= sum_k |A_c[k][n]|: the largest column sums bound both
        double bk = 0.0, bu = 0.0;
        for (int n = 0; n < RV_E; ++n) {
          double cs = 0.0;
          for (int kk = 0; kk < RV_E; ++kk) cs += std::fabs((double)wmp[(size_t)kk * RV_E + n]);
          if (n < RV_U) bk = std::max(bk, cs); else bu = std::max(bu, cs);
        }
        auto pow2_scale = [](double bound) { int ex = 0; if (bound > 0.0 && std::isfinite(bound)) std::frexp(bound * 1.0001, &ex); return std::ldexp(1.0f, 14 - ex); };
        h->mx_kscale = pow2_scale(bk); h->mx_uscale = pow2_scale(bu);
      }
      std::vector<uint16_t> img(RV_WMP16_SLOT);
      float cs[RV_E];
      for (int n = 0; n < RV_E; ++n) {
        float mx = 0.f;
        for (int kk = 0; kk < RV_E; ++kk) mx = std::max(mx, std::fabs(wmp[(size_t)kk * RV_E + n]));
        int ex = 0;
        if (mx > 0.f && std::isfinite(mx)) std::frexp(mx, &ex);
        cs[n] = std::ldexp(1.0f, 14 - ex);
      }
      for (int ks = 0; ks < 8; ++ks)
        for (int nt = 0; nt < 16; ++nt)
          for (int ln = 0; ln < 64; ++ln)
            for (int j = 0; j < 8; ++j) {
              const int n = 16 * nt + (ln & 15);
              const float v = wmp[(size_t)(32 * ks + 8 * (ln >> 4) + j) * RV_E + n] * cs[n];
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (float)hi);
              uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
              img[((((size_t)ks * 16 + nt) * 2 + 0) * 64 + ln) * 8 + j] = hb;
              img[((((size_t)ks * 16 + nt) * 2 + 1) * 64 + ln) * 8 + j] = lb;
            }
      for (int n = 0; n < RV_E; ++n) { const float f = std::ldexp(1.0f, -14) / cs[n]; memcpy(&img[(size_t)2 * RV_E * RV_E + 2 * n], &f, 4); }
      HIPCHK(h, hipMemcpy(h->d_Wmp16, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    if (h->cfg.dec_depth <= 2) {
      // attention = h . A_h + ctx' (h = the top cell's output); its h part is folded into what consumes the attention vector (products in double):
      //   one cell : next cell input  [ctx' | h] . [W_a ; U + A_h W_a],                 logits  ctx' . W_fc + h . (A_h W_fc) + b
      //   two cells: cell 0's input   [ctx' | h_1 | h_0] . [W_a ; A_h W_a ; U_0],       logits  ctx' . W_fc + h_1 . (A_h W_fc) + b
      const int V = h->cfg.vocab;
      const bool two = h->cfg.dec_depth == 2;
      const int KR = two ? 3 * RV_U : RV_E;                 // rows of cell 0's folded kernel
      const size_t wa = (size_t)(h->dec[0].W - h->d_w) + (size_t)V * RV_G, uo = (size_t)(h->dec[0].U - h->d_w);
      const size_t ah = (size_t)(h->W_att - h->d_w), fc = (size_t)(h->W_fc - h->d_w);
      std::vector<float> w2((size_t)KR * RV_G), nh((size_t)RV_U * RV_MAX_VOCAB, 0.f);
      for (int i = 0; i < RV_U; ++i)
        for (int n = 0; n < RV_G; ++n) {
          w2[(size_t)i * RV_G + n] = blob[wa + (size_t)i * RV_G + n];
          double acc = two ? 0.0 : (double)blob[uo + (size_t)i * RV_G + n];
          for (int j = 0; j < RV_U; ++j) acc += (double)blob[ah + (size_t)i * RV_U + j] * (double)blob[wa + (size_t)j * RV_G + n];
          w2[(size_t)(RV_U + i) * RV_G + n] = (float)acc;
          if (two) w2[(size_t)(2 * RV_U + i) * RV_G + n] = blob[uo + (size_t)i * RV_G + n];
        }
      for (int i = 0; i < RV_U; ++i)
        for (int v = 0; v < V; ++v) {
          double acc = 0.0;
          for (int j = 0; j < RV_U; ++j) acc += (double)blob[ah + (size_t)i * RV_U + j] * (double)blob[fc + (size_t)j * V + v];
          nh[(size_t)i * V + v] = (float)acc;
        }
      if (!two) HIPCHK(h, hipMemcpy(h->d_Wcat2, w2.data(), w2.size() * sizeof(float), hipMemcpyHostToDevice));     // (the packed-FMA form of one cell)
      HIPCHK(h, hipMemcpy(h->d_Nh, nh.data(), nh.size() * sizeof(float), hipMemcpyHostToDevice));
      {   // the same kernel for the matrix-pipe cell product (DecState::Wc16): rows divided by the factor their input's f16 image
          // carries (exact: powers of two), one power-of-two scale for the tensor, two f16 parts in B-fragment order per wave
        auto xs = [&](int k) { return k < RV_U ? h->mx_uscale : 16384.f; };
        float mx = 0.f;
        for (int k = 0; k < KR; ++k)
          for (int n = 0; n < RV_G; ++n) mx = std::max(mx, std::fabs(w2[(size_t)k * RV_G + n] / xs(k)));
        int ex = 0;
        if (mx > 0.f && std::isfinite(mx)) std::frexp(mx, &ex);
        const float T = std::ldexp(1.0f, 14 - ex);
        h->mx_cdescale = 1.0f / T;
        const int NP = KR / 8;                              // (k-step, gate) pairs per wave: 32, or 48 with two cells
        std::vector<uint16_t> img((size_t)2 * KR * RV_G);
        for (int wv = 0; wv < 8; ++wv)
          for (int pr = 0; pr < NP; ++pr)
            for (int ln = 0; ln < 64; ++ln)
              for (int j = 0; j < 8; ++j) {
                const int ks = pr >> 2, g = pr & 3, k = 32 * ks + 8 * (ln >> 4) + j, n = RV_U * g + 16 * wv + (ln & 15);
                const float v = (w2[(size_t)k * RV_G + n] / xs(k)) * T;
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
                img[(((((size_t)wv * NP + pr) * 2 + 0) * 64) + ln) * 8 + j] = hb;
                img[(((((size_t)wv * NP + pr) * 2 + 1) * 64) + ln) * 8 + j] = lb;
              }
        HIPCHK(h, hipMemcpy(h->d_Wc16, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        if (two) {   // cell 1's kernels on h_0 (W_1) and on h_1 (U_1): DecState::W1c16, rows divided by the 2^14 their inputs' images carry
          const size_t w1 = (size_t)(h->dec[1].W - h->d_w), u1 = (size_t)(h->dec[1].U - h->d_w);
          float m1 = 0.f;
          for (size_t i = 0; i < (size_t)RV_U * RV_G; ++i) m1 = std::max(m1, std::max(std::fabs(blob[w1 + i]), std::fabs(blob[u1 + i])) / 16384.f);
          int e1 = 0;
          if (m1 > 0.f && std::isfinite(m1)) std::frexp(m1, &e1);
          const float T1 = std::ldexp(1.0f, 14 - e1);
          h->mx_c1descale = 1.0f / T1;
          std::vector<uint16_t> i1((size_t)2 * RV_E * RV_G);
          for (int wv = 0; wv < 8; ++wv)
            for (int pr = 0; pr < 32; ++pr)
              for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                  const int ks = (pr & 15) >> 2, g = pr & 3, k = 32 * ks + 8 * (ln >> 4) + j, n = RV_U * g + 16 * wv + (ln & 15);
                  const float v = (blob[(pr < 16 ? w1 : u1) + (size_t)k * RV_G + n] / 16384.f) * T1;
                  const _Float16 hi = (_Float16)v;
                  const _Float16 lo = (_Float16)(v - (float)hi);
                  uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
                  i1[(((((size_t)wv * 32 + pr) * 2 + 0) * 64) + ln) * 8 + j] = hb;
                  i1[(((((size_t)wv * 32 + pr) * 2 + 1) * 64) + ln) * 8 + j] = lb;
                }
          HIPCHK(h, hipMemcpy(h->d_W1c16, i1.data(), i1.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        // the output layer on the same [ctx' | h] image: logits = ctx' . W_fc + h . (A_h W_fc) + b_fc; columns >= V are zero
        auto wl = [&](int k, int v) -> float {
          if (v >= V) return 0.f;
          return (k < RV_U ? blob[fc + (size_t)k * V + v] : nh[(size_t)(k - RV_U) * V + v]) / xs(k);
        };
        float ml = 0.f;
        for (int k = 0; k < RV_E; ++k)
          for (int v = 0; v < V; ++v) ml = std::max(ml, std::fabs(wl(k, v)));
        int el = 0;
        if (ml > 0.f && std::isfinite(ml)) std::frexp(ml, &el);
        const float Tl = std::ldexp(1.0f, 14 - el);
        h->mx_ldescale = 1.0f / Tl;
        std::vector<uint16_t> li((size_t)2 * RV_E * 16);
        for (int ks = 0; ks < 8; ++ks)
          for (int ln = 0; ln < 64; ++ln)
            for (int j = 0; j < 8; ++j) {
              const float v = wl(32 * ks + 8 * (ln >> 4) + j, ln & 15) * Tl;
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (float)hi);
              uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
              li[(((size_t)ks * 2 + 0) * 64 + ln) * 8 + j] = hb;
              li[(((size_t)ks * 2 + 1) * 64 + ln) * 8 + j] = lb;
            }
        HIPCHK(h, hipMemcpy(h->d_Wl16, li.data(), li.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        if (!two) {   // Bahdanau's query layer W_q [128][128] for the matrix pipe (DecState::Wq16): one power-of-two scale, two f16 parts, per-wave B fragments
          const size_t qo = (size_t)(h->W_q - h->d_w);
          float mq = 0.f;
          for (size_t i = 0; i < (size_t)RV_U * RV_U; ++i) mq = std::max(mq, std::fabs(blob[qo + i]));
          int eq = 0;
          if (mq > 0.f && std::isfinite(mq)) std::frexp(mq, &eq);
          const float Tq = std::ldexp(1.0f, 14 - eq);
          h->mx_qdescale = std::ldexp(1.0f, -14) / Tq;
          std::vector<uint16_t> qi((size_t)2 * RV_U * RV_U);
          for (int wv = 0; wv < 8; ++wv)
            for (int ks = 0; ks < 4; ++ks)
              for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                  const float v = blob[qo + (size_t)(32 * ks + 8 * (ln >> 4) + j) * RV_U + 16 * wv + (ln & 15)] * Tq;
                  const _Float16 hi = (_Float16)v;
                  const _Float16 lo = (_Float16)(v - (float)hi);
                  uint16_t hb, lb; memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
                  qi[((((size_t)wv * 4 + ks) * 2 + 0) * 64 + ln) * 8 + j] = hb;
                  qi[((((size_t)wv * 4 + ks) * 2 + 1) * 64 + ln) * 8 + j] = lb;
                }
          HIPCHK(h, hipMemcpy(h->d_Wq16, qi.data(), qi.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
      }
    }
    const size_t moff = (size_t)(h->W_mem - h->d_w);       // W_mem [256][128] -> [128][256]
    std::vector<float> m((size_t)RV_U * RV_E);
    for (int i = 0; i < RV_E; ++i)
      for (int j = 0; j < RV_U; ++j) m[(size_t)j * RV_E + i] = blob[moff + (size_t)i * RV_U + j];
    HIPCHK(h, hipMemcpy(h->d_WmemT, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->loaded = true;
  return RV_OK;
}

int rv_beam_search(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e,
                   int32_t W, int32_t L, int32_t* tokens, float* scores, int32_t* S_out) {
  return run(h, raw, event, false, B, T_r, T_e, W, L, false, tokens, scores, false, S_out);
}
int rv_beam_search_dev(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e,
                       int32_t W, int32_t L, int32_t* tokens, float* scores, int32_t* S_out) {
  return run(h, raw, event, true, B, T_r, T_e, W, L, false, tokens, scores, true, S_out);
}
int rv_beam_search_calls(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e,
                         int32_t W, int32_t L, const uint8_t* lut, uint8_t* bases, int32_t* lengths, float* probs,
                         int32_t* S_out) {
  const CallsOut c{lut, bases, lengths, probs};
  return run(h, raw, event, false, B, T_r, T_e, W, L, false, nullptr, nullptr, false, S_out, &c);
}
int rv_greedy_search(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e,
                     int32_t L, int32_t* tokens, float* logits, int32_t* S_out) {
  return run(h, raw, event, false, B, T_r, T_e, 1, L, true, tokens, logits, false, S_out);
}
int rv_greedy_search_dev(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e,
                         int32_t L, int32_t* tokens, float* logits, int32_t* S_out) {
  return run(h, raw, event, true, B, T_r, T_e, 1, L, true, tokens, logits, true, S_out);
}

// ---- asynchronous calls: submit returns once the slab's work is queued on one of the handle's contexts; collect waits for it.
static int submit(rv_handle h, const float* raw, const float* ev, bool dev_in, int B, int T_r, int T_e, int W, int L,
                  int32_t* tokens, float* out2, bool dev_out, const uint8_t* lut, int32_t* ticket) {
  if (!h || !ticket) return RV_EINVAL;
  *ticket = -1;
  const int depth = std::min(std::max(h->opt_async_depth, 1), RV_MAX_ASYNC);
  while ((int)h->kids.size() < depth - 1) {
    RvContext* k = nullptr;
    const int rc = create_child(h, &k);
    if (rc != RV_OK) return rc;
    h->kids.push_back(k);
  }
  RvContext* ctx = nullptr; int slot = -1;
  for (int i = 0; i < depth && !ctx; ++i) {          // round robin over the idle contexts
    const int sidx = (h->next_slot + i) % depth;
    RvContext* cand = sidx == 0 ? h : h->kids[sidx - 1];
    if (!cand->pend.busy) { ctx = cand; slot = sidx; }
  }
  if (!ctx) return fail(h, RV_ESTATE, "all %d asynchronous contexts hold uncollected calls (option async_depth): collect one first", depth);
  h->inflight_hint = depth;
  if (ctx != h) sync_child(ctx, h);
  const int rc = enqueue(ctx, raw, ev, dev_in, B, T_r, T_e, W, L, false, tokens, out2, dev_out, lut);
  if (rc != RV_OK) { ctx->pend.busy = false; if (ctx != h) h->err = ctx->err; return rc; }
  h->next_slot = (slot + 1) % depth;
  h->generation = (h->generation + 1) & 0xFFFFF;
  ctx->pend.ticket = (h->generation << 4) | slot;
  *ticket = ctx->pend.ticket;
  return RV_OK;
}
static int collect(rv_handle h, int32_t ticket, int32_t* tokens, float* out2, const CallsOut* calls, int32_t* S_out) {
  if (!h || !S_out) return RV_EINVAL;
  const int slot = ticket & 15;
  if (ticket < 0 || slot > (int)h->kids.size()) return fail(h, RV_EINVAL, "unknown ticket %d", ticket);
  RvContext* ctx = slot == 0 ? h : h->kids[slot - 1];
  if (!ctx->pend.busy || ctx->pend.ticket != ticket) return fail(h, RV_ESTATE, "ticket %d is not in flight (collected already?)", ticket);
  const int rc = finish(ctx, tokens, out2, calls, S_out);
  if (rc != RV_OK && ctx != h) h->err = ctx->err;
  return rc;
}
int rv_beam_search_submit(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e, int32_t W, int32_t L,
                          int32_t* ticket) {
  return submit(h, raw, event, false, B, T_r, T_e, W, L, nullptr, nullptr, false, nullptr, ticket);
}
int rv_beam_search_collect(rv_handle h, int32_t ticket, int32_t* tokens, float* scores, int32_t* S_out) {
  return collect(h, ticket, tokens, scores, nullptr, S_out);
}
int rv_beam_search_submit_dev(rv_handle h, const float* d_raw, const float* d_event, int32_t B, int32_t T_r, int32_t T_e, int32_t W,
                              int32_t L, int32_t* d_tokens, float* d_scores, int32_t* ticket) {
  return submit(h, d_raw, d_event, true, B, T_r, T_e, W, L, d_tokens, d_scores, true, nullptr, ticket);
}
int rv_beam_search_collect_dev(rv_handle h, int32_t ticket, int32_t* S_out) {
  return collect(h, ticket, nullptr, nullptr, nullptr, S_out);
}
int rv_beam_search_submit_calls(rv_handle h, const float* raw, const float* event, int32_t B, int32_t T_r, int32_t T_e, int32_t W,
                                int32_t L, const uint8_t* lut, int32_t* ticket) {
  if (!lut) return h ? fail(h, RV_EINVAL, "null lut") : RV_EINVAL;
  return submit(h, raw, event, false, B, T_r, T_e, W, L, nullptr, nullptr, false, lut, ticket);
}
int rv_beam_search_collect_calls(rv_handle h, int32_t ticket, uint8_t* bases, int32_t* lengths, float* probs, int32_t* S_out) {
  const CallsOut c{nullptr, bases, lengths, probs};
  return collect(h, ticket, nullptr, nullptr, &c, S_out);
}

int rv_set_option(rv_handle h, const char* key, int32_t value) {
  if (!h || !key) return RV_EINVAL;
  h->opt_gen++;                      // slab graphs captured under the old options are rebuilt on their next use
  if (!strcmp(key, "debug_taps")) h->opt_taps = value != 0;
  else if (!strcmp(key, "slab_graph")) h->opt_slab_graph = value != 0;
  else if (!strcmp(key, "persist_taps")) h->opt_ptaps = value != 0;
  else if (!strcmp(key, "use_graph")) h->opt_graph = value != 0;
  else if (!strcmp(key, "flash_attend")) h->opt_flash = value != 0;
  else if (!strcmp(key, "persistent_decode")) h->opt_persist = value != 0;
  else if (!strcmp(key, "concurrent_encoders")) h->opt_side_ev = value != 0;
  else if (!strcmp(key, "lane_projection")) h->opt_lane_inproj = value != 0;
  else if (!strcmp(key, "fused_projection")) h->opt_fuse = value != 0;
  else if (!strcmp(key, "tail_wave")) h->opt_tail_wave = value != 0;
  else if (!strcmp(key, "wide_recurrence")) h->opt_wide = value < 0 ? -1 : (value == 2 ? 2 : (value != 0));
  else if (!strcmp(key, "async_depth")) {
    if (value < 1 || value > RV_MAX_ASYNC) return fail(h, RV_EINVAL, "async_depth must be 1..%d", RV_MAX_ASYNC);
    h->opt_async_depth = value;
    // every context is one HIP stream; the runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the
    // environment says otherwise when the runtime starts).  More contexts than queues: some share a queue and their slabs serialise.
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const int nq = q ? atoi(q) : 4;
    if (value > nq)
      return fail(h, RV_WQUEUES, "async_depth %d set, but GPU_MAX_HW_QUEUES is %s: contexts beyond the hardware queues share one and serialise "
                  "(C3: depth 10 -> 278 k chunks/s on 8 queues, 317 k on 16); set GPU_MAX_HW_QUEUES >= %d in the environment before the "
                  "process's first HIP call", value, q ? q : "unset (4 queues)", value);
  }
  else if (!strcmp(key, "matrix_attention")) h->opt_mx_att = value != 0;
  else if (!strcmp(key, "matrix_cell")) h->opt_mx_cell = value != 0;
  else if (!strcmp(key, "split_projection")) h->opt_split_proj = value < 0 ? 0 : (value > 2 ? 2 : value);
  else if (!strcmp(key, "attend_threads")) {
    if (value != 0 && value != 256 && value != 512) return fail(h, RV_EINVAL, "attend_threads must be 0, 256 or 512");
    h->opt_att_nt = value;
  }
  else if (!strcmp(key, "decode_split")) h->opt_split = value < 1 ? 1 : (value > 4 ? 4 : value);
  else if (!strcmp(key, "profile")) h->opt_profile = value < 0 ? 0 : (value > 3 ? 3 : value);
  else return fail(h, RV_EINVAL, "unknown option '%s'", key);
  return RV_OK;
}

int rv_get_tensor(rv_handle h, const char* name, float* dst, size_t dst_floats, size_t* n_written) {
  if (!h || !name || !n_written) return RV_EINVAL;
  const size_t B = h->lB, W = h->lW, Tm = h->lTm, S = h->lS, V = h->cfg.vocab;
  const void* src = nullptr; size_t n = 0; int kind = 0;   // 0 f32, 1 i32, 2 u8
  const DecState& d = h->dec_st;
  if (!strcmp(name, "enc_output")) { src = h->enc_out; n = B * Tm * RV_E; }
  else if (!strcmp(name, "mask")) { src = h->mask; n = B * Tm; kind = 2; }
  else if (!strcmp(name, "keys")) {
    if (!h->lkeys) return fail(h, RV_ESTATE, "keys were not built by the last call (single-pass attend); set debug_taps=1");
    src = h->keys; n = B * Tm * RV_U;
  }
  else if (!strcmp(name, "projected_memory")) {
    if (!h->lpersist) return fail(h, RV_ESTATE, "the projected memory is built for the persistent decode only (the last call ran the per-step kernels)");
    src = h->mem2; n = B * Tm * RV_E;
  }
  else if (!strcmp(name, "rec_stamps")) {
    if (!h->rec_ts) return fail(h, RV_ESTATE, "set RV_REC_STAMPS=1 before rv_create (and load a -DRV_REC_STAMPS build)");
    long long ts[24];
    HIPCHK(h, hipMemcpy(ts, h->rec_ts, sizeof ts, hipMemcpyDeviceToHost));
    *n_written = 24;
    if (!dst || dst_floats < 24) return fail(h, RV_EINVAL, "rec_stamps needs 24 floats");
    for (int i = 0; i < 24; ++i) dst[i] = (float)ts[i];
    return RV_OK;
  }
  else if (!strcmp(name, "dbg_stamps")) {
    if (!d.dbg_ts) return fail(h, RV_ESTATE, "set RV_DBG_STAMPS=1 before rv_create");
    long long ts[16];
    HIPCHK(h, hipMemcpy(ts, d.dbg_ts, sizeof ts, hipMemcpyDeviceToHost));
    *n_written = 16;
    if (!dst || dst_floats < 16) return fail(h, RV_EINVAL, "dbg_stamps needs 16 floats");
    for (int i = 0; i < 16; ++i) dst[i] = (float)(ts[i] - ts[0]);
    return RV_OK;
  }
  else if ((!strncmp(name, "step_", 5) || !strcmp(name, "parent_ids")) && h->lsplit > 1)
    return fail(h, RV_ESTATE, "per-step records are laid out per sub-slab when decode_split > 1; read them with decode_split=1 or debug_taps=1");
  else if (!strcmp(name, "chunk_steps")) {
    if (!h->lpersist) return fail(h, RV_ESTATE, "chunk_steps exist only after a persistent decode");
    src = h->d_chunk_steps; n = B; kind = 1;
  }
  else if (!strcmp(name, "step_ids")) { src = d.step_ids; n = S * B * W; kind = 1; }
  else if (!strcmp(name, "parent_ids")) { src = d.parent_ids; n = S * B * W; kind = 1; }
  else if (!strcmp(name, "step_scores")) { src = d.step_scores; n = S * B * W; }
  else if (!strcmp(name, "step_logits")) {
    if (!h->ltaps && !h->lgreedy && !h->lptaps) return fail(h, RV_ESTATE, "step_logits needs option debug_taps=1 or persist_taps=1");
    src = d.step_logits; n = S * B * W * V;
  } else if (!strcmp(name, "step_alignments")) {
    if (!h->ltaps) return fail(h, RV_ESTATE, "step_alignments needs option debug_taps=1");
    src = h->step_align; n = S * B * W * Tm;
  } else return fail(h, RV_EINVAL, "unknown tensor '%s'", name);
  *n_written = n;
  if (n == 0) return RV_OK;
  if (!dst || dst_floats < n) return fail(h, RV_EINVAL, "tensor '%s' needs %zu floats, buffer holds %zu", name, n, dst_floats);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (kind == 0) {
    HIPCHK(h, hipMemcpy(dst, src, n * sizeof(float), hipMemcpyDeviceToHost));
    if (!strcmp(name, "step_alignments") && h->lflash) {
      // the single-pass kernel taps raw log2-domain masked scores; normalise here (debug path only)
      for (size_t r = 0; r < n / Tm; ++r) {
        float* a = dst + r * Tm;
        double mx = -INFINITY, sum = 0;
        for (size_t t = 0; t < Tm; ++t) mx = std::max(mx, (double)a[t]);
        for (size_t t = 0; t < Tm; ++t) sum += std::exp2((double)a[t] - mx);
        for (size_t t = 0; t < Tm; ++t) a[t] = (float)(std::exp2((double)a[t] - mx) / sum);
      }
    }
  } else if (kind == 1) {
    std::vector<int32_t> tmp(n);
    HIPCHK(h, hipMemcpy(tmp.data(), src, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) dst[i] = (float)tmp[i];
  } else {
    std::vector<uint8_t> tmp(n);
    HIPCHK(h, hipMemcpy(tmp.data(), src, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) dst[i] = (float)tmp[i];
  }
  return RV_OK;
}

int rv_get_profile(rv_handle h, const char* kernel, double* total_ms, int64_t* launches) {
  if (!h || !kernel || !total_ms || !launches) return RV_EINVAL;
  auto it = h->prof.find(kernel);
  if (it == h->prof.end()) { *total_ms = 0; *launches = 0; return RV_OK; }
  *total_ms = it->second.ms; *launches = it->second.n;
  return RV_OK;
}

int rv_profile_names(rv_handle h, char* dst, size_t dst_bytes) {
  if (!h || !dst || dst_bytes == 0) return RV_EINVAL;
  std::string s;
  for (auto& kv : h->prof) { if (!s.empty()) s += ';'; s += kv.first; }
  if (s.size() + 1 > dst_bytes) return fail(h, RV_EINVAL, "name buffer too small (%zu needed)", s.size() + 1);
  memcpy(dst, s.c_str(), s.size() + 1);
  return RV_OK;
}

int rv_reset_profile(rv_handle h) {
  if (!h) return RV_EINVAL;
  h->prof.clear();
  return RV_OK;
}

}  // extern "C"
