// Native orchestration of the ResNet18 TripletNet(_Finetune)+head graph: forward (eval-folded or train-mode BN),
// backward, gradient all-reduce (RCCL, bucketed + overlapped), synced BatchNorm, fused optimizer.  One C call per
// step; every kernel goes to the caller's stream.  See include/sslcr.h for the reference code each entry replaces.
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "kernels.hpp"

using namespace sslcr;

#define TRY(expr)                                   \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return check(_e, #expr);  \
  } while (0)
#define TRYI(expr)            \
  do {                        \
    int _r = (expr);          \
    if (_r != 0) return _r;   \
  } while (0)
#define TRYN(expr)                                                                     \
  do {                                                                                 \
    ncclResult_t _r = (expr);                                                          \
    if (_r != ncclSuccess) return fail("%s: %s", #expr, ncclGetErrorString(_r));       \
  } while (0)

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) {
      hipError_t e = hipFree(p);          // synchronises the device: only ever happens while shapes grow
      if (e != hipSuccess) return check(e, "hipFree");
      p = nullptr;
      cap = 0;
    }
    bytes = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return check(e, "hipMalloc");
    cap = bytes;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Carver {                 // bump allocator over a DevBuf region (offsets only; ensure() afterwards)
  size_t off = 0;
  size_t take(size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

struct ConvL {
  int cin = 0, cout = 0, k = 0, stride = 1, pad = 0, pidx = -1;
  void *w_fwd = nullptr, *w_dg = nullptr, *w_fold = nullptr;
  float* b_fold = nullptr;
  // bf16 storage: the eval pack keeps the PLAIN filter and the BatchNorm scale goes to the conv's epilogue (sslcr_conv_desc.out_scale):
  // the loss error of the full-size iteration is 3.6 x smaller than with the scale folded in before the rounding; nullptr (fp32
  // mode): folded into w_fold
  float* s_fold = nullptr;
  void* w_foldf = nullptr;        // stem only, with s_fold: the FOLDED eval pack for the shapes the fused conv + max-pool kernel does not take
                                  // (stem_fwd_kernel has no output scale: its registers, see launch_stem)
  // fp8 engine mode (SSLCR_FP8): e4m3 shadow packs + per-kout dequant factors of the train / eval-folded filters, for the
  // convs the fp8 kernel serves (3x3 stride 1, cin and cout multiples of 128: layers 2-4)
  uint8_t *w8_fwd = nullptr, *w8_fold = nullptr;
  float *dq_fwd = nullptr, *dq_fold = nullptr;
  float* f8s = nullptr;     // delayed activation scaling: {x_scale, amax} of the train forward, {x_scale, amax} of the eval forward
};
struct BnL {
  int C = 0, pg = -1, pb = -1, bidx = -1;
};
struct BlockL {
  ConvL c1, c2, ds;
  BnL b1, b2, bd;
  bool has_ds = false;
  int pstart = 0;
};
struct BnSaved {
  float *scale, *shift, *mean, *invstd;
};
struct PassState {
  const void* x = nullptr;
  const void* x2 = nullptr;      // second input segment (images n >= n_split), see sslcr_stem_desc
  int n_split = 0;
  int in_f32 = 0, N = 0, H = 0, W = 0;
  DevBuf mem;
  char* raw0 = nullptr;
  char* pooled = nullptr;
  uint8_t* argmax = nullptr;
  struct {
    char *raw1, *raw2, *rawd, *y;
    uint8_t* ybits;              // bf16 mode: one bit per element of y, (y > 0) -- what bn2's backward reduce pass reads instead of y
  } blk[8];
  BnSaved bn[20];
  float* E = nullptr;
};

}  // namespace

struct ProfRec {
  hipEvent_t e0, e1;
  double flops, bytes;
  const char* name;       // the kernel template instance that was launched
};
struct Profiler {               // optional HIP-event bracketing of the conv launches (bench.py roofline leg)
  bool on = false;
  std::vector<ProfRec> rec[3];  // 0: conv (fwd + dgrad), 1: wgrad, 2: bn_bwd_apply (the largest HBM-bound kernel; flops = 0)
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
};

// "Virtual ranks": W engine contexts of ONE process on ONE device (one host thread and one stream each) exchange through this
// object instead of RCCL -- every sharded code path of the engine (synced BatchNorm forward/backward sums, global-count loss
// scaling, bucketed gradient sums on the side stream) then runs with world = W on a single-GPU box and can be compared with the
// 1-rank step on the concatenated batch (tests/test_engine_gpu2.py).  RCCL refuses two ranks on one device, hence the stand-in;
// it is deterministic (ranks summed in rank order) and asynchronous on the callers' streams like the real collective.
constexpr int VW_MAX = 8;
struct VChan {                      // one per collective stream (0: BatchNorm sums on the compute streams, 1: gradient buckets)
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  long gen = 0;
  void* slot[2][VW_MAX] = {};       // [call parity][rank]: that rank's contribution, device memory owned by the rank
  size_t cap[2][VW_MAX] = {};
  hipEvent_t ev_in[2][VW_MAX] = {}, ev_out[2][VW_MAX] = {};
  bool out_set[2][VW_MAX] = {};
  long calls[VW_MAX] = {};
};
struct sslcr_vcomm {
  int world = 1;
  VChan ch[2];
  std::atomic<bool> poisoned{false};   // a rank timed out inside a collective: its call parity is out of step with its peers for good
};

struct sslcr_ctx {
  Profiler prof;
  int device = 0, dtype = 0;
  int fp8 = 0;                    // SSLCR_FP8: dtype stays bf16 (storage, backward); eligible forward convs run the e4m3 kernel
  sslcr_vcomm* vcomm = nullptr;   // set instead of comm / comm_g by sslcr_comm_init_virtual
  ncclComm_t comm = nullptr;      // BatchNorm-sum all-reduces, only ever used on the caller's (compute) stream
  ncclComm_t comm_g = nullptr;    // gradient buckets, only ever used on comm_stream (one communicator per stream, like
                                  // separate process groups: no cross-stream serialisation inside RCCL)
  int rank = 0, world = 1;
  int bn_sync = 1;                // train-mode BatchNorm on GLOBAL batch statistics (all-reduce of per-channel sums); 0 = per replica
  hipStream_t comm_stream = nullptr;
  // second compute stream: the (frozen, eval-mode) teacher forward of an SSL_CR step runs next to the student forward, so the
  // workgroups of one fill the tail rounds of the other's persistent kernels (the two share no buffers: the teacher works in
  // ctx scratch, which the student only touches in backward)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_aux_begin = nullptr, ev_aux_end = nullptr;
  int use_aux = 0;                // sslcr_set_aux_stream
  // weight-gradient side stream: in backward the wgrad launches (MFMA-bound, 3 ms of the step) depend only on a BatchNorm-backward
  // output and feed nothing but the gradient buffer, while the chain they leave behind alternates MFMA-bound dgrads with
  // HBM-bound BatchNorm passes (2.4 ms) that leave the matrix pipes idle -- so they are launched on a second stream behind an
  // event and fill those gaps.  kinds: 0 = a block's conv2 (reads the dRaw2 scratch), 1 = conv1 (dRaw1), 2 = projection (dRawD);
  // ev_wg_done[k] orders the NEXT writer of that scratch buffer behind the reader.
  hipStream_t wg_stream = nullptr;
  hipEvent_t ev_wg_in[3] = {nullptr, nullptr, nullptr}, ev_wg_done[3] = {nullptr, nullptr, nullptr}, ev_wg_join = nullptr;
  bool wg_pending[3] = {false, false, false};
  bool wg_any = false;            // something was launched on wg_stream since the last join
  int fuse_stem_bwd = 1;          // conv1 wgrad derives dY from the pooled gradient in LDS (SSLCR_FUSE_STEM_BWD=0: apply pass + wgrad, for A/B runs)
  int use_wg = 0;                 // sslcr_set_wgrad_stream (default off: measured neutral, see include/sslcr.h; off while profiling)
  hipEvent_t ev_ready[8], ev_done = nullptr;
  DevBuf scratch;     // eval-forward activations and backward transients (never live at the same time)
  DevBuf partials;    // BN partial rows
  DevBuf small;       // bn stage/sums, unit scale/shift
  double* bn_stage = nullptr;
  double* bn_sums = nullptr;
  int* bn_tickets = nullptr;      // sslcr_bn_finalize_desc.tickets: 16 zeroed ints (every BatchNorm finalize of this context runs on one stream at a time)
  // BatchNorm-backward sums: every reduce pass of one backward takes its own [2][2][512]-double slot of this ring (the pass
  // overwrites it with the ordered sum of its workgroups' rows: nothing to clear)
  DevBuf bn_ring;
  int bn_ring_i = 0;
  static constexpr int kBnRing = 96, kBnSlot = 2 * 2 * 512;
  float *ones = nullptr, *zeros = nullptr;
  size_t esz() const { return dtype == DT_BF16 ? 2 : 4; }
};

struct sslcr_net {
  sslcr_ctx* ctx = nullptr;
  const void* split_x2 = nullptr;   // set by sslcr_step_ssl_cr around the student pass: input = (x[0:split_n], split_x2)
  int split_n = 0;
  int nparams = 0, head_kind = 0, ncls = 1, triplet = 0;
  std::vector<float*> params;
  std::vector<uint8_t> rg;
  std::vector<int> psize;
  std::vector<size_t> goff;
  float* bn_rm[20];
  float* bn_rv[20];
  int64_t* bn_nbt[20];
  ConvL stem;
  BnL bn0;
  BlockL blocks[8];
  DevBuf shadow, grads, heads, descs, chunks, f8buf;
  float* f8slots = nullptr;       // [n8][2] = {x_scale, amax since the last update}: two slots (train, eval) per fp8 conv, see ConvL::f8s
  int n8 = 0;
  bool f8_cal[2] = {false, false};  // the first forward of this net in (train, eval) mode has calibrated the activation scales
  bool f8_calib_pass = false;       // a calibration pass is running: no running-statistics update, outputs discarded
  int nchunks = 0;
  bool opt_packs_all = false;     // the optimizer work list rewrites every non-stem conv's train-mode shadow weights
  size_t grad_count = 0;
  PassState pass[3];
  bool ybits_ok = false;         // the passes' mask bits are allocated back to back (what the segment forms assume)
  // heads state (fp32)
  int hN = 0;
  float *cat[3], *hact[3], *fi[3], *feats = nullptr, *hid = nullptr, *logits = nullptr, *dE[3], *dfeats = nullptr, *dhid = nullptr,
        *dtmp256 = nullptr, *dtmp512 = nullptr, *dcat = nullptr, *scratch = nullptr, *dlogits = nullptr, *logits_t = nullptr;
  int last_N = 0, last_npass = 0;
  bool last_segments = false;      // the last train forward ran the branches as segments
  bool packed_train = false, packed_eval = false;
  // sslcr_net_debug_tap: backward keeps copies of each block's transient gradient tensors (pass 0) for the layer-wise replay test
  bool tap = false;
  DevBuf tapbuf[8][6];
  int tapdims[8][6][4] = {};
  int tap_flags[8] = {};            // bit 0: dAct1 was written with bn1's ReLU mask already applied (mask_x front end of the dgrad)
  std::vector<sslcr_tensor_desc> host_descs;
  std::vector<float*> st1, st2;
  int ndesc = 0, max_n = 0;
};

namespace {

struct VSrc { const void* p[VW_MAX]; };
template <typename T>
__global__ void vsum_kernel(T* dst, VSrc src, int W, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T s = static_cast<const T*>(src.p[0])[i];
    for (int q = 1; q < W; ++q) s += static_cast<const T*>(src.p[q])[i];       // rank order: every rank gets the same bits
    dst[i] = s;
  }
}

std::atomic<int> g_live_ctx{0};     // contexts between sslcr_create and sslcr_destroy (see sslcr_destroy)

inline bool sharded(const sslcr_ctx* c) { return c->comm != nullptr || c->vcomm != nullptr; }

int vcomm_all_reduce(sslcr_ctx* c, int chan, void* buf, size_t count, bool f64, hipStream_t st) {
  sslcr_vcomm* v = c->vcomm;
  VChan& ch = v->ch[chan];
  const int r = c->rank, W = v->world;
  const size_t bytes = count * (f64 ? 8 : 4);
  if (v->poisoned) return fail("virtual all-reduce: the communicator was poisoned by an earlier timeout");
  const int ph = (int)(ch.calls[r]++ & 1);
  // the slot is reused every second call: its previous readers (the peers' sum kernels of two calls ago) recorded ev_out then
  for (int q = 0; q < W; ++q)
    if (q != r && ch.out_set[ph][q]) TRY(hipStreamWaitEvent(st, ch.ev_out[ph][q], 0));
  if (ch.cap[ph][r] < bytes) {
    if (ch.slot[ph][r]) TRY(hipFree(ch.slot[ph][r]));       // (synchronises the device)
    ch.slot[ph][r] = nullptr; ch.cap[ph][r] = 0;
    TRY(hipMalloc(&ch.slot[ph][r], (bytes + 255) & ~(size_t)255));
    ch.cap[ph][r] = (bytes + 255) & ~(size_t)255;
  }
  if (!ch.ev_in[ph][r]) {
    TRY(hipEventCreateWithFlags(&ch.ev_in[ph][r], hipEventDisableTiming));
    TRY(hipEventCreateWithFlags(&ch.ev_out[ph][r], hipEventDisableTiming));
  }
  TRY(hipMemcpyAsync(ch.slot[ph][r], buf, bytes, hipMemcpyDeviceToDevice, st));
  TRY(hipEventRecord(ch.ev_in[ph][r], st));
  {                                                   // host rendezvous: every rank has published its slot and event
    std::unique_lock<std::mutex> lk(ch.m);
    const long g = ch.gen;
    if (++ch.arrived == W) {
      ch.arrived = 0; ++ch.gen;
      ch.cv.notify_all();
    } else if (!ch.cv.wait_for(lk, std::chrono::seconds(120), [&] { return ch.gen != g; })) {
      --ch.arrived;
      c->vcomm->poisoned = true;          // calls[r] and the slot are already published: every later collective would pair up wrongly
      ch.cv.notify_all();
      return fail("virtual all-reduce: rank %d waited 120 s for its peers (channel %d); the communicator is now unusable", r, chan);
    }
  }
  VSrc src;
  for (int q = 0; q < W; ++q) {
    src.p[q] = ch.slot[ph][q];
    if (q != r) TRY(hipStreamWaitEvent(st, ch.ev_in[ph][q], 0));
  }
  const int blocks = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
  if (f64) hipLaunchKernelGGL(vsum_kernel<double>, dim3(blocks), dim3(256), 0, st, (double*)buf, src, W, count);
  else hipLaunchKernelGGL(vsum_kernel<float>, dim3(blocks), dim3(256), 0, st, (float*)buf, src, W, count);
  TRY(hipGetLastError());
  TRY(hipEventRecord(ch.ev_out[ph][r], st));
  ch.out_set[ph][r] = true;
  return 0;
}

// SUM all-reduce in place over the ranks of this job: channel 0 = BatchNorm sums (compute stream), 1 = gradient buckets (side stream)
int all_reduce(sslcr_ctx* c, int chan, void* buf, size_t count, bool f64, hipStream_t st) {
  if (c->vcomm) return vcomm_all_reduce(c, chan, buf, count, f64, st);
  TRYN(ncclAllReduce(buf, buf, count, f64 ? ncclDouble : ncclFloat, ncclSum, chan == 0 ? c->comm : c->comm_g, st));
  return 0;
}

// ---- launch wrappers: when profiling is on, bracket the kernel with HIP events on ITS stream and book the
// algorithmic work (forward-conv FLOPs even for the strided dgrad gather, whose zero taps are not counted)
hipError_t prof_conv(sslcr_ctx* c, int dt, const ConvArgs& a, hipStream_t st) {
  if (!c->prof.on) return launch_conv(dt, a, st);
  ProfRec r;
  r.e0 = c->prof.get(); r.e1 = c->prof.get();
  const double es = c->esz();
  const double rs = a.tap_mask ? (double)__builtin_popcount(a.tap_mask) : (double)a.R * a.S;
  const double M = (double)a.N * a.PH * a.PW;
  const double src = (double)a.N * a.H * a.W;
  r.flops = a.transposed ? 2.0 * src * a.C * a.K * rs : 2.0 * M * a.K * a.C * rs;
  r.bytes = (src * a.C + (double)a.K * rs * a.C + M * a.K * (a.residual ? 2.0 : 1.0) * (a.accumulate ? 2.0 : 1.0)) * es;
  r.name = conv_kernel_name(dt, a);
  (void)hipEventRecord(r.e0, st);
  hipError_t e = launch_conv(dt, a, st);
  (void)hipEventRecord(r.e1, st);
  c->prof.rec[0].push_back(r);
  return e;
}
// a downsampling block's conv1 (3x3 / 2) and 1x1 / 2 projection of one input in one launch (conv_s2.hip)
hipError_t prof_conv_pair(sslcr_ctx* c, const ConvArgs& a, const ConvArgs& d, hipStream_t st) {
  if (!c->prof.on) return launch_conv_s2(a, &d, st);
  ProfRec r;
  r.e0 = c->prof.get(); r.e1 = c->prof.get();
  const double es = c->esz();
  const double M = (double)a.N * a.PH * a.PW, src = (double)a.N * a.H * a.W;
  r.flops = 2.0 * M * a.K * a.C * 10.0;
  r.bytes = (src * a.C + (double)a.K * 10.0 * a.C + 2.0 * M * a.K) * es;
  r.name = conv_s2_name(a, true);
  (void)hipEventRecord(r.e0, st);
  hipError_t e = launch_conv_s2(a, &d, st);
  (void)hipEventRecord(r.e1, st);
  c->prof.rec[0].push_back(r);
  return e;
}
inline bool fp8_layer(const sslcr_ctx* c, const ConvL& L) {
  return c->fp8 && L.k == 3 && L.stride == 1 && L.cin % 128 == 0 && L.cout % 128 == 0;
}
// forward conv of layer L (folded = eval pack): the fp8 kernel where the mode, the layer and the shape allow it, else the
// engine dtype's kernel
inline bool use_fp8(const sslcr_ctx* c, const ConvL& L, const ConvArgs& a) { return fp8_layer(c, L) && L.w8_fwd && conv_fp8_mode(a) != 0; }
hipError_t prof_conv_fwd(sslcr_ctx* c, int dt, const ConvArgs& a_in, const ConvL& L, bool folded, hipStream_t st) {
  // (the e4m3 eval pack is the FOLDED filter with its own per-kout dequant factor: no output scale on that path)
  ConvArgs a = a_in;
  a.out_scale = nullptr;
  if (!use_fp8(c, L, a)) return prof_conv(c, dt, a_in, st);
  Fp8Args q;
  q.w8 = folded ? L.w8_fold : L.w8_fwd;
  q.w_dequant = folded ? L.dq_fold : L.dq_fwd;
  // per-tensor activation scale by DELAYED scaling: this launch records the amax of what it quantises, the next forward of
  // this net in this mode uses 2^floor(log2(448 / (2 amax))) (scale 1 until then); no host sync anywhere
  q.x_scale = 1.0f;
  q.x_scale_dev = L.f8s ? L.f8s + (folded ? 2 : 0) : nullptr;
  q.amax_out = L.f8s ? L.f8s + (folded ? 3 : 1) : nullptr;
  if (!c->prof.on) return launch_conv_fp8(a, q, st);
  ProfRec r;
  r.e0 = c->prof.get(); r.e1 = c->prof.get();
  const double M = (double)a.N * a.PH * a.PW;
  r.flops = 2.0 * M * a.K * a.C * 9.0;
  r.bytes = ((double)a.N * a.H * a.W * a.C + M * a.K * (a.residual ? 2.0 : 1.0)) * 2.0 + (double)a.K * 9.0 * a.C;
  r.name = conv_fp8_name(a);
  (void)hipEventRecord(r.e0, st);
  hipError_t e = launch_conv_fp8(a, q, st);
  (void)hipEventRecord(r.e1, st);
  c->prof.rec[0].push_back(r);
  return e;
}
hipError_t prof_wgrad(sslcr_ctx* c, int dt, const WgradArgs& a, hipStream_t st) {
  if (!c->prof.on) return launch_wgrad(dt, a, st);
  ProfRec r;
  r.e0 = c->prof.get(); r.e1 = c->prof.get();
  const double es = c->esz();
  const double M = (double)a.N * a.OH * a.OW;
  r.flops = 2.0 * M * a.K * a.C * a.R * a.S;
  r.bytes = ((double)a.N * a.H * a.W * a.C + M * a.K) * es + (double)a.K * a.R * a.S * a.C * 4.0;
  r.name = wgrad_kernel_name(dt, a);
  (void)hipEventRecord(r.e0, st);
  hipError_t e = launch_wgrad(dt, a, st);
  (void)hipEventRecord(r.e1, st);
  c->prof.rec[1].push_back(r);
  return e;
}

constexpr int kMaxSeg = 3;      // TripletNet branches run as segments of one launch (backbone_forward_train_segments)
const int kBlockCfg[8][3] ={{64, 64, 1}, {64, 64, 1}, {64, 128, 2}, {128, 128, 1}, {128, 256, 2}, {256, 256, 1}, {256, 512, 2}, {512, 512, 1}};

inline int out_dim(int h, int k, int stride, int pad) { return (h + 2 * pad - k) / stride + 1; }

void build_topology(sslcr_net* n) {
  int p = 0, b = 0;
  n->stem = ConvL{3, 64, 7, 2, 3, p++};
  n->bn0 = BnL{64, p, p + 1, b++};
  p += 2;
  for (int i = 0; i < 8; ++i) {
    BlockL& B = n->blocks[i];
    const int cin = kBlockCfg[i][0], cout = kBlockCfg[i][1], s = kBlockCfg[i][2];
    B.pstart = p;
    B.c1 = ConvL{cin, cout, 3, s, 1, p++};
    B.b1 = BnL{cout, p, p + 1, b++};
    p += 2;
    B.c2 = ConvL{cout, cout, 3, 1, 1, p++};
    B.b2 = BnL{cout, p, p + 1, b++};
    p += 2;
    B.has_ds = (s != 1 || cin != cout);
    if (B.has_ds) {
      B.ds = ConvL{cin, cout, 1, s, 0, p++};
      B.bd = BnL{cout, p, p + 1, b++};
      p += 2;
    }
  }
  // p == 60 ; heads: 60 fc.0.weight [512,1024], 61 fc.0.bias, 62 fc.2.weight [256,512], 63 fc.2.bias, 64.. classifier
  n->psize.assign(n->nparams, 0);
  auto conv_sz = [](const ConvL& c) { return c.cout * c.cin * c.k * c.k; };
  n->psize[n->stem.pidx] = conv_sz(n->stem);
  n->psize[n->bn0.pg] = n->psize[n->bn0.pb] = 64;
  for (int i = 0; i < 8; ++i) {
    BlockL& B = n->blocks[i];
    n->psize[B.c1.pidx] = conv_sz(B.c1);
    n->psize[B.c2.pidx] = conv_sz(B.c2);
    n->psize[B.b1.pg] = n->psize[B.b1.pb] = B.b1.C;
    n->psize[B.b2.pg] = n->psize[B.b2.pb] = B.b2.C;
    if (B.has_ds) {
      n->psize[B.ds.pidx] = conv_sz(B.ds);
      n->psize[B.bd.pg] = n->psize[B.bd.pb] = B.bd.C;
    }
  }
  n->psize[60] = 512 * 1024;
  n->psize[61] = 512;
  n->psize[62] = 256 * 512;
  n->psize[63] = 256;
  if (n->head_kind == 0) {
    n->psize[64] = n->ncls * 768;
    n->psize[65] = n->ncls;
  } else {
    n->psize[64] = 128 * 768;
    n->psize[65] = 128;
    n->psize[66] = n->ncls * 128;
    n->psize[67] = n->ncls;
  }
  n->goff.assign(n->nparams + 1, 0);
  for (int i = 0; i < n->nparams; ++i) n->goff[i + 1] = n->goff[i] + ((n->psize[i] + 63) & ~63);
  n->grad_count = n->goff[n->nparams];
}

// bf16 storage: eval-mode BatchNorm is NOT folded into the filters (the scale rides in the conv epilogues).  SSLCR_UNFOLD_EVAL=0 folds as
// rounds 1-5 did (A/B runs, tools/bf16_teacher_fold_experiment.py)
bool unfold_eval(const sslcr_ctx* c) {
  static const bool on = [] { const char* e = getenv("SSLCR_UNFOLD_EVAL"); return !e || atoi(e) != 0; }();
  return on && c->dtype == DT_BF16;
}

int alloc_shadow(sslcr_net* n) {
  const size_t es = n->ctx->esz();
  Carver c;
  std::vector<std::pair<ConvL*, size_t>> offs;
  auto add = [&](ConvL& L) {
    const size_t wbytes = (L.pidx == 0 ? (size_t)64 * 224 : (size_t)L.cout * L.cin * L.k * L.k) * es;
    size_t o = c.take(wbytes);          // w_fwd
    c.take(wbytes);                     // w_dg
    c.take(wbytes);                     // w_fold
    c.take(L.cout * sizeof(float));     // b_fold
    c.take(L.cout * sizeof(float));     // s_fold
    if (L.pidx == 0) c.take(wbytes);    // w_foldf
    if (fp8_layer(n->ctx, L)) {         // w8_fwd, w8_fold, dq_fwd, dq_fold
      c.take((size_t)L.cout * 9 * L.cin); c.take((size_t)L.cout * 9 * L.cin);
      c.take(L.cout * sizeof(float)); c.take(L.cout * sizeof(float));
    }
    offs.push_back({&L, o});
  };
  add(n->stem);
  for (int i = 0; i < 8; ++i) {
    add(n->blocks[i].c1);
    add(n->blocks[i].c2);
    if (n->blocks[i].has_ds) add(n->blocks[i].ds);
  }
  TRYI(n->shadow.ensure(c.off));
  for (auto& pr : offs) {
    ConvL& L = *pr.first;
    const size_t wbytes = ((L.pidx == 0 ? (size_t)64 * 224 : (size_t)L.cout * L.cin * L.k * L.k) * es + 255) & ~(size_t)255;
    char* base = (char*)n->shadow.p + pr.second;
    L.w_fwd = base;
    L.w_dg = base + wbytes;
    L.w_fold = base + 2 * wbytes;
    L.b_fold = (float*)(base + 3 * wbytes);
    {
      const size_t bb = (L.cout * sizeof(float) + 255) & ~(size_t)255;
      L.s_fold = unfold_eval(n->ctx) ? (float*)(base + 3 * wbytes + bb) : nullptr;
      if (L.pidx == 0) L.w_foldf = base + 3 * wbytes + 2 * bb;
    }
    if (fp8_layer(n->ctx, L)) {
      const size_t bb = (L.cout * sizeof(float) + 255) & ~(size_t)255, w8 = ((size_t)L.cout * 9 * L.cin + 255) & ~(size_t)255;
      char* q = base + 3 * wbytes + 2 * bb;
      L.w8_fwd = (uint8_t*)q; L.w8_fold = (uint8_t*)(q + w8);
      L.dq_fwd = (float*)(q + 2 * w8); L.dq_fold = (float*)(q + 2 * w8 + bb);
    }
  }
  if (n->ctx->fp8) {
    n->n8 = 0;
    for (auto& pr : offs) if (pr.first->w8_fwd) n->n8 += 2;
    TRYI(n->f8buf.ensure((size_t)n->n8 * 2 * sizeof(float)));
    n->f8slots = (float*)n->f8buf.p;
    std::vector<float> init((size_t)n->n8 * 2);
    for (int i = 0; i < n->n8; ++i) { init[2 * i] = 1.f; init[2 * i + 1] = 0.f; }
    TRY(hipMemcpy(n->f8slots, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice));
    int k = 0;
    for (auto& pr : offs) if (pr.first->w8_fwd) { pr.first->f8s = n->f8slots + 4 * k; ++k; }
  }
  return 0;
}

int pack_conv_layer(sslcr_net* n, ConvL& L, const BnL& bn, int mode, hipStream_t st) {
  const int dt = n->ctx->dtype;
  PackArgs a;
  memset(&a, 0, sizeof(a));
  a.w = n->params[L.pidx];
  a.K = L.cout; a.C = L.cin; a.R = L.k; a.S = L.k;
  a.eps = 1e-5f;
  const bool stem = (L.pidx == 0);
  if (mode & 1) {
    a.w_fwd = L.w_fwd;
    a.w_dgrad = stem ? nullptr : L.w_dg;
    a.dgrad_flip = (L.k == 3 && L.stride == 1);     // stride-1 dgrad runs as a plain 3x3 conv of dY (halo kernel)
    TRY(stem ? launch_pack_stem(dt, a, st) : launch_pack_conv(dt, a, st));
  }
  if (mode & 2) {
    a.w_fwd = L.w_fold;
    a.w_dgrad = nullptr;
    a.gamma = n->params[bn.pg]; a.beta = n->params[bn.pb];
    a.rmean = n->bn_rm[bn.bidx]; a.rvar = n->bn_rv[bn.bidx];
    a.bias_out = L.b_fold;
    a.scale_out = L.s_fold;                          // (non-null: w_fold is the plain filter, the scale goes to the epilogue)
    TRY(stem ? launch_pack_stem(dt, a, st) : launch_pack_conv(dt, a, st));
    if (stem && L.s_fold && L.w_foldf) {             // ... and the folded form for the stem's two-kernel path
      a.w_fwd = L.w_foldf; a.scale_out = nullptr;
      TRY(launch_pack_stem(dt, a, st));
    }
  }
  if (L.w8_fwd) {
    PackFp8Args f;
    memset(&f, 0, sizeof(f));
    f.w = n->params[L.pidx]; f.K = L.cout; f.C = L.cin; f.eps = 1e-5f;
    if (mode & 1) {
      f.w8 = L.w8_fwd; f.w_dequant = L.dq_fwd;
      TRY(launch_pack_fp8(f, st));
    }
    if (mode & 2) {
      f.w8 = L.w8_fold; f.w_dequant = L.dq_fold;
      f.gamma = n->params[bn.pg]; f.beta = n->params[bn.pb]; f.rmean = n->bn_rm[bn.bidx]; f.rvar = n->bn_rv[bn.bidx];
      TRY(launch_pack_fp8(f, st));        // (the folded bias is the bf16 pack's b_fold)
    }
  }
  return 0;
}

// ---------------------------------------------------------------- BN finalize (optionally synced across ranks)
// nseg > 1: `rows` covers nseg segments (equal shares, in order), sv is segment 0's and the others follow seg_stride floats apart
int finalize_bn(sslcr_net* n, const BnL& bn, const float* partials, int rows, double local_count, BnSaved& sv, int replay, hipStream_t st,
                int nseg = 1, int seg_stride = 0) {
  sslcr_ctx* c = n->ctx;
  BnFinalizeArgs a;
  memset(&a, 0, sizeof(a));
  a.partials = partials; a.rows = rows; a.C = bn.C;
  a.nseg = nseg; a.seg_stride = seg_stride;
  if (nseg > 1 && sharded(c) && c->bn_sync) return fail("finalize_bn: segments with synced BatchNorm");
  a.gamma = n->params[bn.pg]; a.beta = n->params[bn.pb];
  a.scale = sv.scale; a.shift = sv.shift; a.mean = sv.mean; a.invstd = sv.invstd;
  a.running_mean = n->bn_rm[bn.bidx]; a.running_var = n->bn_rv[bn.bidx]; a.num_batches_tracked = n->bn_nbt[bn.bidx];
  if (n->f8_calib_pass) { a.running_mean = nullptr; a.running_var = nullptr; a.num_batches_tracked = nullptr; }
  a.momentum = 0.1f; a.eps = 1e-5f; a.replay = replay;
  a.stage = c->bn_stage;
  a.tickets = c->bn_tickets;
  if (sharded(c) && c->bn_sync) {
    // global-batch statistics: reduce rows -> [2][C] sums, all-reduce, finalize from the sums
    BnFinalizeArgs r = a;
    r.sums_out = c->bn_sums;
    TRY(launch_bn_finalize(r, st));
    TRYI(all_reduce(c, 0, c->bn_sums, 2 * (size_t)bn.C, true, st));
    a.sums_in = c->bn_sums;
    a.count = local_count * c->world;
  } else {
    a.count = local_count;
  }
  TRY(launch_bn_finalize(a, st));
  return 0;
}

// a downsampling block's bn1 and projection BatchNorm, whose statistics rows come out of ONE launch (the stride-2 pair): with synced
// BatchNorm their sums share one all-reduce ([2][C] each, adjacent in bn_sums) -- 3 of the step's 37 latency-bound collectives less
int finalize_bn_pair(sslcr_net* n, const BnL& b1, const float* part1, int rows1, const BnL& bd, const float* partd, int rowsd, double local_count,
                     BnSaved& sv1, BnSaved& svd, int replay, hipStream_t st, int nseg, int seg_stride) {
  sslcr_ctx* c = n->ctx;
  if (!(sharded(c) && c->bn_sync) || b1.C != bd.C || nseg > 1) {
    TRYI(finalize_bn(n, b1, part1, rows1, local_count, sv1, replay, st, nseg, seg_stride));
    return finalize_bn(n, bd, partd, rowsd, local_count, svd, replay, st, nseg, seg_stride);
  }
  const BnL* bns[2] = {&b1, &bd};
  const float* parts[2] = {part1, partd};
  const int rowsv[2] = {rows1, rowsd};
  BnSaved* svs[2] = {&sv1, &svd};
  BnFinalizeArgs a[2];
  for (int i = 0; i < 2; ++i) {
    memset(&a[i], 0, sizeof(a[i]));
    a[i].partials = parts[i]; a[i].rows = rowsv[i]; a[i].C = bns[i]->C;
    a[i].gamma = n->params[bns[i]->pg]; a[i].beta = n->params[bns[i]->pb];
    a[i].scale = svs[i]->scale; a[i].shift = svs[i]->shift; a[i].mean = svs[i]->mean; a[i].invstd = svs[i]->invstd;
    a[i].running_mean = n->bn_rm[bns[i]->bidx]; a[i].running_var = n->bn_rv[bns[i]->bidx]; a[i].num_batches_tracked = n->bn_nbt[bns[i]->bidx];
    if (n->f8_calib_pass) { a[i].running_mean = nullptr; a[i].running_var = nullptr; a[i].num_batches_tracked = nullptr; }
    a[i].momentum = 0.1f; a[i].eps = 1e-5f; a[i].replay = replay;
    a[i].stage = c->bn_stage; a[i].tickets = c->bn_tickets;
    BnFinalizeArgs r = a[i];
    r.sums_out = c->bn_sums + (size_t)i * 2 * b1.C;
    TRY(launch_bn_finalize(r, st));                  // rows -> this rank's sums (the stage is consumed before the next launch reuses it)
  }
  TRYI(all_reduce(c, 0, c->bn_sums, 4 * (size_t)b1.C, true, st));
  for (int i = 0; i < 2; ++i) {
    a[i].sums_in = c->bn_sums + (size_t)i * 2 * b1.C;
    a[i].count = local_count * c->world;
    TRY(launch_bn_finalize(a[i], st));
  }
  return 0;
}

ConvArgs conv_args(const ConvL& L, const void* x, const void* w, void* y, int N, int H, int W) {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.y = y;
  a.N = N; a.H = H; a.W = W; a.C = L.cin; a.K = L.cout; a.R = L.k; a.S = L.k; a.stride = L.stride; a.pad = L.pad;
  a.PH = out_dim(H, L.k, L.stride, L.pad); a.PW = out_dim(W, L.k, L.stride, L.pad);
  a.OH = a.PH; a.OW = a.PW; a.osh = 1;
  return a;
}

int ensure_partials(sslcr_ctx* c, const ConvArgs& a, float** out, int* rows, bool fp8 = false) {
  *rows = fp8 ? conv_fp8_rows(a) : conv_partials_rows(a);
  TRYI(c->partials.ensure((size_t)*rows * 2 * a.K * sizeof(float)));
  *out = (float*)c->partials.p;
  return 0;
}

struct Dims {
  int oh0, ow0, ph, pw;
  int lh[8], lw[8];     // OUTPUT spatial dims of each block
};
Dims make_dims(int H, int W) {
  Dims d;
  d.oh0 = out_dim(H, 7, 2, 3); d.ow0 = out_dim(W, 7, 2, 3);
  d.ph = out_dim(d.oh0, 3, 2, 1); d.pw = out_dim(d.ow0, 3, 2, 1);
  int h = d.ph, w = d.pw;
  for (int i = 0; i < 8; ++i) {
    const int s = kBlockCfg[i][2];
    h = out_dim(h, 3, s, 1); w = out_dim(w, 3, s, 1);
    d.lh[i] = h; d.lw[i] = w;
  }
  return d;
}

// Saved activations of ALL passes (1, or the 3 TripletNet branches) in one buffer, laid out [tensor][pass][...]: a tensor of
// pass p+1 follows the same tensor of pass p, so a weight-gradient launch can take the branches as one 3N-image batch
// (dW is linear in the pixels; at N=128 per branch the per-launch fp32 atomics into dW dominated the RSP step).
int alloc_passes(sslcr_net* n, int npass, int N, int H, int W) {
  const size_t es = n->ctx->esz();
  const Dims d = make_dims(H, W);
  Carver c;
  auto take = [&](size_t per_pass) { return c.take(per_pass * npass); };   // (every per-pass size is a multiple of 128 bytes)
  const size_t s_raw0 = (size_t)N * d.oh0 * d.ow0 * 64 * es, s_pool = (size_t)N * d.ph * d.pw * 64 * es, s_arg = (size_t)N * d.ph * d.pw * 64;
  const size_t o_raw0 = take(s_raw0), o_pool = take(s_pool), o_arg = take(s_arg);
  size_t o_blk[8][4], s_blk[8], o_bits[8], s_bits[8];
  bool bits_ok = true;
  for (int i = 0; i < 8; ++i) {
    s_blk[i] = (size_t)N * d.lh[i] * d.lw[i] * kBlockCfg[i][1] * es;
    for (int j = 0; j < 4; ++j) o_blk[i][j] = (j == 2 && !n->blocks[i].has_ds) ? 0 : take(s_blk[i]);
    s_bits[i] = (((size_t)N * d.lh[i] * d.lw[i] * kBlockCfg[i][1] / 8) + 127) / 128 * 128;
    o_bits[i] = take(s_bits[i]);
    if (s_bits[i] * 8 != (size_t)N * d.lh[i] * d.lw[i] * kBlockCfg[i][1]) bits_ok = false;      // padded: not contiguous across passes
  }
  const size_t s_bn = 20 * 4 * 512 * sizeof(float), s_E = (size_t)N * 512 * sizeof(float);
  const size_t o_bn = take(s_bn), o_E = take(s_E);
  TRYI(n->pass[0].mem.ensure(c.off));
  char* b = (char*)n->pass[0].mem.p;
  for (int p = 0; p < npass; ++p) {
    PassState& ps = n->pass[p];
    ps.raw0 = b + o_raw0 + p * s_raw0; ps.pooled = b + o_pool + p * s_pool; ps.argmax = (uint8_t*)(b + o_arg + p * s_arg);
    for (int i = 0; i < 8; ++i) {
      ps.blk[i].raw1 = b + o_blk[i][0] + p * s_blk[i]; ps.blk[i].raw2 = b + o_blk[i][1] + p * s_blk[i];
      ps.blk[i].rawd = n->blocks[i].has_ds ? b + o_blk[i][2] + p * s_blk[i] : nullptr;
      ps.blk[i].y = b + o_blk[i][3] + p * s_blk[i];
      ps.blk[i].ybits = (uint8_t*)(b + o_bits[i] + p * s_bits[i]);
    }
    float* f = (float*)(b + o_bn + p * s_bn);
    for (int i = 0; i < 20; ++i) {
      ps.bn[i] = BnSaved{f, f + 512, f + 1024, f + 1536};
      f += 2048;
    }
    ps.E = (float*)(b + o_E + p * s_E);
    ps.N = N; ps.H = H; ps.W = W;
  }
  n->ybits_ok = bits_ok;
  return 0;
}

// ---------------------------------------------------------------- backbone forward, train mode (saves everything)
// stem of one pass: conv1 7x7/2 with statistics -> bn0 finalize -> BatchNorm + ReLU + max-pool into ps.pooled
int forward_stem(sslcr_net* n, PassState& ps, const void* x, int in_f32, int N, int H, int W, int replay, hipStream_t st) {
  sslcr_ctx* c = n->ctx;
  const int dt = c->dtype;
  const Dims d = make_dims(H, W);
  ps.x = x; ps.in_f32 = in_f32;
  StemArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x2 = ps.x2; a.n_split = ps.n_split; a.w = n->stem.w_fwd; a.y = ps.raw0;
  a.N = N; a.H = H; a.W = W; a.OH = d.oh0; a.OW = d.ow0; a.in_f32 = in_f32;
  const int rows = stem_partials_rows(a);
  TRYI(c->partials.ensure((size_t)rows * 2 * 64 * sizeof(float)));
  a.stats = (float*)c->partials.p;
  TRY(launch_stem(dt, a, st));
  TRYI(finalize_bn(n, n->bn0, a.stats, rows, (double)N * d.oh0 * d.ow0, ps.bn[0], replay, st));
  PoolFwdArgs p;
  memset(&p, 0, sizeof(p));
  p.x = ps.raw0; p.scale = ps.bn[0].scale; p.shift = ps.bn[0].shift; p.y = ps.pooled; p.argmax = ps.argmax;
  p.N = N; p.H = d.oh0; p.W = d.ow0; p.C = 64; p.OH = d.ph; p.OW = d.pw;
  TRY(launch_bn_relu_maxpool(dt, p, st));
  return 0;
}

// The ReLU mask of a block output as bits (sslcr_bn_act_desc.ybits, bf16 mode): bn2's backward reduce pass reads 1/16 of the bytes of y.
// SSLCR_YBITS=0 keeps the tensor read (same-box A/B runs).
bool ybits_on(const sslcr_ctx* c) {
  static const bool on = [] { const char* e = getenv("SSLCR_YBITS"); return !e || atoi(e) != 0; }();
  return on && c->dtype == DT_BF16;
}
// the bits that go with a saved block output (nullptr: not one of them)
const uint8_t* ybits_of(const sslcr_net* n, const void* yact) {
  if (!yact || !n->ybits_ok || !ybits_on(n->ctx)) return nullptr;
  for (int p = 0; p < 3; ++p)
    for (int i = 0; i < 8; ++i)
      if (n->pass[p].blk[i].y == yact) return n->pass[p].blk[i].ybits;
  return nullptr;
}

// the eight residual blocks + average pool from ps.pooled on.  nseg == 1: one pass of N images.  nseg > 1: ps is the FIRST of nseg
// passes whose saved tensors follow one another (alloc_passes) -- every layer is then one launch over nseg * N images with
// seg_images = N in its descriptor, one finalize and one bn_act for the nseg BatchNorm batches (see segments_servable)
int forward_blocks(sslcr_net* n, PassState& ps, int N, int H, int W, int replay, int nseg, hipStream_t st) {
  sslcr_ctx* c = n->ctx;
  const int dt = c->dtype;
  const Dims d = make_dims(H, W);
  const bool segs = nseg > 1;
  const int NT = nseg * N;
  const int seg_stride = segs ? (int)(n->pass[1].bn[0].scale - n->pass[0].bn[0].scale) : 0;
  const char* X = ps.pooled;
  int xh = d.ph, xw = d.pw;
  for (int i = 0; i < 8; ++i) {
    BlockL& B = n->blocks[i];
    const int oh = d.lh[i], ow = d.lw[i];
    const double cnt = (double)N * oh * ow;
    float* part; int rows;
    ConvArgs a1 = conv_args(B.c1, X, B.c1.w_fwd, ps.blk[i].raw1, NT, xh, xw);
    if (segs) a1.seg_images = N;
    // a downsampling block's conv1 and projection read the same input: one launch where the plane-gather kernel serves the pair
    // (both sets of statistics rows, the projection's BatchNorm finalized before conv2 reuses the rows buffer)
    ConvArgs ad;
    bool paired = false;
    if (B.has_ds) {
      ad = conv_args(B.ds, X, B.ds.w_fwd, ps.blk[i].rawd, NT, xh, xw);
      if (segs) ad.seg_images = N;
      ConvArgs t1 = a1, td = ad;
      t1.stats = td.stats = c->zeros;                // (any non-null pointer: the predicate looks at the mode only)
      paired = !use_fp8(c, B.c1, a1) && conv_s2_pair_ok(dt, t1, td);
    }
    if (paired) {
      if (segs && (!conv_segments_ok(dt, a1) || !conv_segments_ok(dt, ad))) return fail("forward_blocks: the paired stride-2 launch has no segment form here");
      rows = conv_partials_rows(a1);
      const size_t rb = (size_t)rows * 2 * a1.K * sizeof(float);
      TRYI(c->partials.ensure(2 * rb));
      part = (float*)c->partials.p;
      a1.stats = part;
      ad.stats = (float*)((char*)c->partials.p + rb);
      TRY(prof_conv_pair(c, a1, ad, st));
      TRYI(finalize_bn_pair(n, B.b1, a1.stats, rows, B.bd, ad.stats, rows, cnt, ps.bn[B.b1.bidx], ps.bn[B.bd.bidx], replay, st, nseg, seg_stride));
    } else if (B.has_ds) {
      // two launches (fp32, layer4.0's 8x8 maps, ragged shapes), the projection right behind conv1 so that the two BatchNorms still
      // share one all-reduce of their sums when BatchNorm is synced across ranks
      const int rows1 = conv_partials_rows(a1), rowsd = conv_partials_rows(ad);
      const size_t rb1 = (((size_t)rows1 * 2 * a1.K * sizeof(float)) + 255) & ~(size_t)255;
      TRYI(c->partials.ensure(rb1 + (size_t)rowsd * 2 * ad.K * sizeof(float)));
      a1.stats = (float*)c->partials.p;
      ad.stats = (float*)((char*)c->partials.p + rb1);
      TRY(prof_conv(c, dt, a1, st));
      TRY(prof_conv(c, dt, ad, st));
      TRYI(finalize_bn_pair(n, B.b1, a1.stats, rows1, B.bd, ad.stats, rowsd, cnt, ps.bn[B.b1.bidx], ps.bn[B.bd.bidx], replay, st, nseg, seg_stride));
      paired = true;                                   // (the projection has run: the block's tail below must not launch it again)
    } else {
      TRYI(ensure_partials(c, a1, &part, &rows, !segs && use_fp8(c, B.c1, a1)));
      a1.stats = part;
      TRY(segs ? prof_conv(c, dt, a1, st) : prof_conv_fwd(c, dt, a1, B.c1, false, st));
      TRYI(finalize_bn(n, B.b1, part, rows, cnt, ps.bn[B.b1.bidx], replay, st, nseg, seg_stride));
    }
    ConvArgs a2 = conv_args(B.c2, ps.blk[i].raw1, B.c2.w_fwd, ps.blk[i].raw2, NT, oh, ow);
    a2.in_scale = ps.bn[B.b1.bidx].scale; a2.in_shift = ps.bn[B.b1.bidx].shift; a2.in_relu = 1;
    if (segs) { a2.seg_images = N; a2.seg_stride = seg_stride; }
    TRYI(ensure_partials(c, a2, &part, &rows, !segs && use_fp8(c, B.c2, a2)));
    a2.stats = part;
    TRY(segs ? prof_conv(c, dt, a2, st) : prof_conv_fwd(c, dt, a2, B.c2, false, st));
    TRYI(finalize_bn(n, B.b2, part, rows, cnt, ps.bn[B.b2.bidx], replay, st, nseg, seg_stride));
    BnActArgs e;
    memset(&e, 0, sizeof(e));
    e.x = ps.blk[i].raw2; e.scale = ps.bn[B.b2.bidx].scale; e.shift = ps.bn[B.b2.bidx].shift;
    e.y = ps.blk[i].y; e.pixels = (size_t)NT * oh * ow; e.C = B.c2.cout; e.relu = 1;
    if (segs) { e.nseg = nseg; e.seg_stride = seg_stride; }
    if (n->ybits_ok && ybits_on(c)) e.ybits = ps.blk[i].ybits;
    if (B.has_ds) {
      if (!paired) {
        TRYI(ensure_partials(c, ad, &part, &rows));
        ad.stats = part;
        TRY(prof_conv(c, dt, ad, st));
        TRYI(finalize_bn(n, B.bd, part, rows, cnt, ps.bn[B.bd.bidx], replay, st, nseg, seg_stride));
      }
      e.res = ps.blk[i].rawd; e.rscale = ps.bn[B.bd.bidx].scale; e.rshift = ps.bn[B.bd.bidx].shift;
    } else {
      e.res = X;
    }
    TRY(launch_bn_act(dt, e, st));
    X = ps.blk[i].y; xh = oh; xw = ow;
  }
  TRY(launch_avgpool_fwd(dt, X, ps.E, NT, xh * xw, 512, st));
  return 0;
}

int backbone_forward_train(sslcr_net* n, PassState& ps, const void* x, int in_f32, int N, int H, int W, int replay, hipStream_t st) {
  if (n->pass[0].N != N || n->pass[0].H != H || n->pass[0].W != W || !n->pass[0].mem.p) TRYI(alloc_passes(n, n->triplet ? 3 : 1, N, H, W));
  ps.x2 = n->split_x2; ps.n_split = n->split_x2 ? n->split_n : 0;
  TRYI(forward_stem(n, ps, x, in_f32, N, H, W, replay, st));
  return forward_blocks(n, ps, N, H, W, replay, 1, st);
}

// The TripletNet branches (models/net.py:50-66: three tiles through ONE backbone, BatchNorm statistics per call) as SEGMENTS of one
// launch per layer: the saved tensors of the passes are contiguous (alloc_passes), so a conv over 3N images with
// sslcr_conv_desc.seg_images = N writes all three, its statistics rows split by segment, and one finalize / one bn_act serves the
// three BatchNorm batches (running statistics updated in branch order).  At N = 128 per branch the per-pass launches were 60 %
// slower per image than the same kernels at N = 640 (r03: conv3x3_h16 48.7 us against 149 us for five times the images).
// The stem (three separate input tensors) and its pool stay per pass.  false = some layer has no segment form here: the caller
// runs the passes one by one.
bool segments_on() {
  static const bool on = [] { const char* e = getenv("SSLCR_SEGMENTS"); return !(e && e[0] == '0'); }();
  return on;
}
bool segments_servable(sslcr_net* n, int N, int H, int W) {
  sslcr_ctx* c = n->ctx;
  if (!segments_on() || !n->triplet || c->dtype != DT_BF16 || c->fp8 || (sharded(c) && c->bn_sync) || c->prof.on) return false;
  const Dims d = make_dims(H, W);
  int xh = d.ph, xw = d.pw;
  for (int i = 0; i < 8; ++i) {
    BlockL& B = n->blocks[i];
    const int oh = d.lh[i], ow = d.lw[i];
    ConvArgs a1 = conv_args(B.c1, nullptr, nullptr, nullptr, 3 * N, xh, xw);
    a1.seg_images = N;
    ConvArgs a2 = conv_args(B.c2, nullptr, nullptr, nullptr, 3 * N, oh, ow);
    a2.seg_images = N; a2.in_scale = c->ones; a2.in_shift = c->zeros;
    if (!conv_segments_ok(c->dtype, a1) || !conv_segments_ok(c->dtype, a2)) return false;
    if (B.has_ds) {
      ConvArgs ad = conv_args(B.ds, nullptr, nullptr, nullptr, 3 * N, xh, xw);
      ad.seg_images = N;
      if (!conv_segments_ok(c->dtype, ad)) return false;
    }
    if (conv_partials_rows(a1) % 3 || conv_partials_rows(a2) % 3) return false;
    xh = oh; xw = ow;
  }
  return true;
}

int backbone_forward_train_segments(sslcr_net* n, const void* const* xs, int in_f32, int N, int H, int W, hipStream_t st) {
  constexpr int NS = 3;
  if (n->pass[0].N != N || n->pass[0].H != H || n->pass[0].W != W || !n->pass[0].mem.p) TRYI(alloc_passes(n, NS, N, H, W));
  for (int p = 0; p < NS; ++p) {
    n->pass[p].x2 = nullptr; n->pass[p].n_split = 0;
    TRYI(forward_stem(n, n->pass[p], xs[p], in_f32, N, H, W, 1, st));
  }
  return forward_blocks(n, n->pass[0], N, H, W, 1, NS, st);
}

// ---------------------------------------------------------------- backbone forward, eval mode (BN folded, nothing saved)
// npass > 1 (TripletNet under model.eval(): validate() of the RSP scripts): nothing in this mode depends on the batch, so the
// branches' stems write into one buffer and every block conv runs once over npass * N images
int backbone_forward_eval(sslcr_net* n, const void* const* xs, int npass, int in_f32, int N, int H, int W, float* const* E, hipStream_t st) {
  sslcr_ctx* c = n->ctx;
  const int dt = c->dtype;
  const size_t es = c->esz();
  const Dims d = make_dims(H, W);
  const int NT = npass * N;
  Carver cv;
  const size_t o_a0 = cv.take((size_t)N * d.oh0 * d.ow0 * 64 * es);
  const size_t unit1 = (size_t)N * d.ph * d.pw * 64 * es, unit = unit1 * npass;
  size_t o_buf[4];
  for (int j = 0; j < 4; ++j) o_buf[j] = cv.take(unit);
  TRYI(c->scratch.ensure(cv.off));
  char* base = (char*)c->scratch.p;
  for (int p = 0; p < npass; ++p) {
    char* pooled = base + o_buf[0] + p * unit1;
    StemArgs a;
    memset(&a, 0, sizeof(a));
    a.x = xs[p]; a.w = n->stem.w_fold; a.y = base + o_a0; a.bias = n->stem.b_fold; a.relu = 1; a.out_scale = n->stem.s_fold;
    a.N = N; a.H = H; a.W = W; a.OH = d.oh0; a.OW = d.ow0; a.in_f32 = in_f32;
    if (stem_pool_ok(dt, a, d.ph, d.pw)) {
      a.y = pooled;                                        // conv1 + folded BatchNorm + ReLU + max-pool in one launch: the conv output stays on the CU
      TRY(launch_stem_pool(dt, a, d.ph, d.pw, st));
    } else {
      if (a.out_scale) { a.w = n->stem.w_foldf; a.out_scale = nullptr; }      // (the two-kernel path takes the folded pack)
      TRY(launch_stem(dt, a, st));
      PoolFwdArgs q;
      memset(&q, 0, sizeof(q));
      q.x = base + o_a0; q.y = pooled;                     // plain max-pool: the stem's epilogue applied the folded BatchNorm + ReLU
      q.N = N; q.H = d.oh0; q.W = d.ow0; q.C = 64; q.OH = d.ph; q.OW = d.pw;
      TRY(launch_bn_relu_maxpool(dt, q, st));
    }
  }
  int xi = 0, xh = d.ph, xw = d.pw;
  for (int i = 0; i < 8; ++i) {
    BlockL& B = n->blocks[i];
    char* X = base + o_buf[xi];
    char* t1 = base + o_buf[(xi + 1) & 3];
    char* td = base + o_buf[(xi + 2) & 3];
    char* Y = base + o_buf[(xi + 3) & 3];
    const int oh = d.lh[i], ow = d.lw[i];
    ConvArgs a1 = conv_args(B.c1, X, B.c1.w_fold, t1, NT, xh, xw);
    a1.bias = B.c1.b_fold; a1.relu = 1; a1.out_scale = B.c1.s_fold;
    const void* res = X;
    if (B.has_ds) {
      ConvArgs ad = conv_args(B.ds, X, B.ds.w_fold, td, NT, xh, xw);
      ad.bias = B.ds.b_fold; ad.out_scale = B.ds.s_fold;
      if (!use_fp8(c, B.c1, a1) && conv_s2_pair_ok(dt, a1, ad)) {      // conv1 and the projection in one launch
        TRY(prof_conv_pair(c, a1, ad, st));
      } else {
        TRY(prof_conv_fwd(c, dt, a1, B.c1, true, st));
        TRY(prof_conv(c, dt,ad, st));
      }
      res = td;
    } else {
      TRY(prof_conv_fwd(c, dt, a1, B.c1, true, st));
    }
    ConvArgs a2 = conv_args(B.c2, t1, B.c2.w_fold, Y, NT, oh, ow);
    a2.bias = B.c2.b_fold; a2.residual = res; a2.relu = 1; a2.out_scale = B.c2.s_fold;
    TRY(prof_conv_fwd(c, dt, a2, B.c2, true, st));
    xi = (xi + 3) & 3; xh = oh; xw = ow;
  }
  for (int p = 0; p < npass; ++p)
    TRY(launch_avgpool_fwd(dt, base + o_buf[xi] + (size_t)p * N * xh * xw * 512 * es, E[p], N, xh * xw, 512, st));
  return 0;
}

// ---------------------------------------------------------------- heads
int alloc_heads(sslcr_net* n, int N) {
  if (n->hN >= N && n->heads.p) return 0;
  Carver c;
  size_t o_cat[3], o_h[3], o_f[3], o_dE[3];
  for (int i = 0; i < 3; ++i) {
    o_cat[i] = c.take((size_t)N * 1024 * 4); o_h[i] = c.take((size_t)N * 512 * 4);
    o_f[i] = c.take((size_t)N * 256 * 4); o_dE[i] = c.take((size_t)N * 512 * 4);
  }
  const size_t o_feats = c.take((size_t)N * 768 * 4), o_hid = c.take((size_t)N * 128 * 4), o_log = c.take((size_t)N * 64 * 4);
  const size_t o_dfeats = c.take((size_t)N * 768 * 4), o_dhid = c.take((size_t)N * 128 * 4);
  const size_t o_d256 = c.take((size_t)N * 256 * 4), o_d512 = c.take((size_t)N * 512 * 4), o_dcat = c.take((size_t)N * 1024 * 4);
  const size_t o_scr = c.take((size_t)N * 1024 * 4), o_dl = c.take((size_t)N * 64 * 4), o_lt = c.take((size_t)N * 64 * 4);
  TRYI(n->heads.ensure(c.off));
  char* b = (char*)n->heads.p;
  for (int i = 0; i < 3; ++i) {
    n->cat[i] = (float*)(b + o_cat[i]); n->hact[i] = (float*)(b + o_h[i]);
    n->fi[i] = (float*)(b + o_f[i]); n->dE[i] = (float*)(b + o_dE[i]);
  }
  n->feats = (float*)(b + o_feats); n->hid = (float*)(b + o_hid); n->logits = (float*)(b + o_log);
  n->dfeats = (float*)(b + o_dfeats); n->dhid = (float*)(b + o_dhid);
  n->dtmp256 = (float*)(b + o_d256); n->dtmp512 = (float*)(b + o_d512); n->dcat = (float*)(b + o_dcat);
  n->scratch = (float*)(b + o_scr); n->dlogits = (float*)(b + o_dl); n->logits_t = (float*)(b + o_lt);
  n->hN = N;
  return 0;
}

const int kPairs[3][2] = {{0, 1}, {1, 2}, {0, 2}};     // E12, E23, E13 (models/net.py:56-58)

// E[0..npass-1] -> feats [N,768], logits [N,C]
int heads_forward(sslcr_net* n, float* const* E, int npass, int N, hipStream_t st) {
  const float *w0 = n->params[60], *b0 = n->params[61], *w2 = n->params[62], *b2 = n->params[63];
  const int nh = (npass == 3) ? 3 : 1;
  for (int i = 0; i < nh; ++i) {
    const float* Ea = E[npass == 3 ? kPairs[i][0] : 0];
    const float* Eb = E[npass == 3 ? kPairs[i][1] : 0];
    if (Ea == Eb) {
      TRY(launch_copy2d_multi(n->cat[i], 1024, 2, 512, Ea, 512, 1, 0, N, 512, st));
    } else {
      TRY(launch_copy2d(n->cat[i], 1024, Ea, 512, N, 512, 0, st));
      TRY(launch_copy2d(n->cat[i] + 512, 1024, Eb, 512, N, 512, 0, st));
    }
    TRY(launch_linear_fwd(n->cat[i], w0, b0, n->hact[i], N, 512, 1024, 1, st));
    TRY(launch_linear_fwd(n->hact[i], w2, b2, n->fi[i], N, 256, 512, 0, st));
  }
  if (nh == 1) {
    TRY(launch_copy2d_multi(n->feats, 768, 3, 256, n->fi[0], 256, 1, 0, N, 256, st));
  } else {
    for (int i = 0; i < 3; ++i) TRY(launch_copy2d(n->feats + 256 * i, 768, n->fi[i], 256, N, 256, 0, st));
  }
  if (n->head_kind == 0) {
    TRY(launch_linear_fwd(n->feats, n->params[64], n->params[65], n->logits, N, n->ncls, 768, 0, st));
  } else {
    TRY(launch_linear_fwd(n->feats, n->params[64], n->params[65], n->hid, N, 128, 768, 1, st));
    TRY(launch_linear_fwd(n->hid, n->params[66], n->params[67], n->logits, N, n->ncls, 128, 0, st));
  }
  return 0;
}

inline float* gptr(sslcr_net* n, int pidx) { return n->rg[pidx] ? (float*)n->grads.p + n->goff[pidx] : nullptr; }

// dlogits -> head parameter grads and dE[0..npass-1]
int heads_backward(sslcr_net* n, const float* dlogits, int npass, int N, bool need_dE, hipStream_t st) {
  const float *w0 = n->params[60], *w2 = n->params[62];
  const bool any_fc = n->rg[60] || n->rg[61] || n->rg[62] || n->rg[63];
  if (n->head_kind == 0) {
    TRY(launch_linear_bwd(n->feats, n->params[64], dlogits, nullptr, (any_fc || need_dE) ? n->dfeats : nullptr, gptr(n, 64), gptr(n, 65),
                          N, n->ncls, 768, 0, n->scratch, st));
  } else {
    TRY(launch_linear_bwd(n->hid, n->params[66], dlogits, nullptr, n->dhid, gptr(n, 66), gptr(n, 67), N, n->ncls, 128, 0, n->scratch, st));
    TRY(launch_linear_bwd(n->feats, n->params[64], n->dhid, n->hid, (any_fc || need_dE) ? n->dfeats : nullptr, gptr(n, 64), gptr(n, 65),
                          N, 128, 768, 0, n->scratch, st));
  }
  if (!any_fc && !need_dE) return 0;
  const int nh = (npass == 3) ? 3 : 1;
  if (need_dE && nh == 3)
    for (int i = 0; i < npass; ++i) TRY(hipMemsetAsync(n->dE[i], 0, (size_t)N * 512 * 4, st));
  for (int i = 0; i < nh; ++i) {
    if (nh == 3) {
      TRY(launch_copy2d(n->dtmp256, 256, n->dfeats + 256 * i, 768, N, 256, 0, st));
    } else {   // the three identical branches of TripletNet_Finetune: gradients add (models/net.py:96-100)
      TRY(launch_copy2d_multi(n->dtmp256, 256, 1, 0, n->dfeats, 768, 3, 256, N, 256, st));
    }
    TRY(launch_linear_bwd(n->hact[i], w2, n->dtmp256, nullptr, n->dtmp512, gptr(n, 62), gptr(n, 63), N, 256, 512, 0, n->scratch, st));
    TRY(launch_linear_bwd(n->cat[i], w0, n->dtmp512, n->hact[i], need_dE ? n->dcat : nullptr, gptr(n, 60), gptr(n, 61), N, 512, 1024, 0,
                          n->scratch, st));
    if (need_dE && nh == 1) {                    // both halves of [E, E] are the one embedding: dE = left + right
      TRY(launch_copy2d_multi(n->dE[0], 512, 1, 0, n->dcat, 1024, 2, 512, N, 512, st));
    } else if (need_dE) {
      const int a = nh == 3 ? kPairs[i][0] : 0, b = nh == 3 ? kPairs[i][1] : 0;
      TRY(launch_copy2d(n->dE[a], 512, n->dcat, 1024, N, 512, 1, st));
      TRY(launch_copy2d(n->dE[b], 512, n->dcat + 512, 1024, N, 512, 1, st));
    }
  }
  return 0;
}

// ---------------------------------------------------------------- backbone backward for one saved pass
struct PoolSrc {            // gradient arriving through the stem max-pool (see sslcr_bn_bwd_desc.pool_dy)
  const void* dy; const uint8_t* argmax; int H, W, OH, OW; const void* y;
};

// BatchNorm backward in three parts so that independent BatchNorms (a block's bn2 and its projection-shortcut BatchNorm) can
// share ONE all-reduce of their sums: begin = the reduce pass into `sums` ([2][C] doubles), sync = the all-reduce, end = the apply pass
// the next sums slot of this backward (nullptr when the ring is used up: the caller then uses c->bn_sums)
double* take_sums(sslcr_ctx* c) {
  if (!c->bn_ring.p || c->bn_ring_i >= sslcr_ctx::kBnRing) return nullptr;
  return (double*)c->bn_ring.p + (size_t)(c->bn_ring_i++) * sslcr_ctx::kBnSlot;
}

// nseg consecutive slots (nullptr if the ring cannot serve them)
double* take_sums_n(sslcr_ctx* c, int nseg) {
  if (!c->bn_ring.p || c->bn_ring_i + nseg > sslcr_ctx::kBnRing) return nullptr;
  double* p = (double*)c->bn_ring.p + (size_t)c->bn_ring_i * sslcr_ctx::kBnSlot;
  c->bn_ring_i += nseg;
  return p;
}

int bn_bwd_begin(sslcr_net* n, const BnL& bn, const BnSaved& sv, const void* dy, const void* x, const void* yact, int relu_from_x,
                 void* dx, void* gout, size_t pixels, double count, hipStream_t st, double* sums, BnBwdArgs* out, const PoolSrc* pool = nullptr,
                 int g_in_reduce = 0, const float* sum_rows = nullptr, int n_sum_rows = 0, bool sums_zeroed = false, int nseg = 1,
                 int seg_stride = 0, bool defer = false) {
  // nseg > 1 (sslcr_bn_bwd_desc.nseg): the tensors hold nseg passes one after the other, `pixels` is their total, sv is pass 0's
  // (the others seg_stride floats apart), `sums` the first of nseg consecutive ring slots, sum_rows / n_sum_rows cover all passes
  sslcr_ctx* c = n->ctx;
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.x = x; a.yact = yact; a.scale = sv.scale; a.shift = sv.shift; a.mean = sv.mean; a.invstd = sv.invstd;
  a.sums = sums; a.dx = dx; a.gout = gout; a.pixels = pixels; a.C = bn.C; a.relu_from_x = relu_from_x;
  if (nseg > 1) { a.nseg = nseg; a.seg_stride = seg_stride; a.sums_stride = sslcr_ctx::kBnSlot; }
  a.g_in_reduce = (g_in_reduce && yact && gout) ? 1 : 0;
  a.yact_bits = ybits_of(n, yact);
  const bool synced = sharded(c) && c->bn_sync;
  if (n->rg[bn.pg] || n->rg[bn.pb]) {
    // dgamma/dbeta ride on the apply pass.  Synced BN: every rank holds the GLOBAL sums and the gradient all-reduce adds
    // `world` copies -> pre-divide (per-replica BN: the sums are this rank's share, the gradient all-reduce adds them up)
    a.dgamma = gptr(n, bn.pg); a.dbeta = gptr(n, bn.pb);
    a.pg_scale = synced ? 1.0f / c->world : 1.0f;
    if (!a.dgamma || !a.dbeta) { a.dgamma = nullptr; a.dbeta = nullptr; }
  }
  if (pool) { a.pool_dy = pool->dy; a.pool_argmax = pool->argmax; a.pH = pool->H; a.pW = pool->W; a.pOH = pool->OH; a.pOW = pool->OW; a.pool_y = pool->y; }
  a.count = synced ? count * c->world : count;
  if (sum_rows) {
    // the dgrad that produced dy already left partial rows of (sum g, sum g (x - mean)) (sslcr_conv_desc.mask_x): rows -> sums
    BnFinalizeArgs r;
    memset(&r, 0, sizeof(r));
    r.partials = sum_rows; r.rows = n_sum_rows; r.C = bn.C; r.stage = c->bn_stage; r.sums_out = sums; r.tickets = c->bn_tickets;
    if (nseg > 1) { r.nseg = nseg; r.seg_stride = sslcr_ctx::kBnSlot; }
    TRY(launch_bn_finalize(r, st));
  } else if (!defer) {                              // (defer: the caller launches the reduce pass itself -- the two-BatchNorm form)
    (void)sums_zeroed;                              // (the reduce pass overwrites its sums)
    TRY(launch_bn_bwd_reduce(c->dtype, a, st));
  }
  *out = a;
  return 0;
}

int bn_bwd_sync(sslcr_ctx* c, double* sums, size_t count, hipStream_t st) {
  if (sharded(c) && c->bn_sync) TRYI(all_reduce(c, 0, sums, count, true, st));
  return 0;
}

int bn_bwd_end(sslcr_ctx* c, const BnBwdArgs& a, hipStream_t st) {
  if (c->prof.on) {
    ProfRec r;
    r.e0 = c->prof.get(); r.e1 = c->prof.get();
    const bool pool = a.pool_dy != nullptr;
    if (pool) r.name = c->dtype == DT_BF16 ? "sslcr::bn_bwd_apply_pool_kernel<unsigned short>" : "sslcr::bn_bwd_apply_pool_kernel<float>";
    else r.name = c->dtype == DT_BF16 ? "sslcr::bn_bwd_apply_kernel<unsigned short>" : "sslcr::bn_bwd_apply_kernel<float>";
    r.flops = 0.0;
    // algorithmic bytes: read dy (or the 4x smaller pooled gradient + 1-byte argmax), x, (saved output for the ReLU mask); write dx (, g)
    const double t = (double)a.pixels * a.C * c->esz();
    const double rd = (pool ? 0.25 * t + 0.25 * (double)a.pixels * a.C : t) + t + ((a.yact && !a.g_in_reduce) ? t : 0.0);
    r.bytes = rd + t + ((a.gout && !a.g_in_reduce) ? t : 0.0);
    (void)hipEventRecord(r.e0, st);
    hipError_t e = launch_bn_bwd_apply(c->dtype, a, st);
    (void)hipEventRecord(r.e1, st);
    c->prof.rec[2].push_back(r);
    TRY(e);
  } else {
    TRY(launch_bn_bwd_apply(c->dtype, a, st));
  }
  return 0;
}

int bn_backward(sslcr_net* n, const BnL& bn, const BnSaved& sv, const void* dy, const void* x, const void* yact, int relu_from_x,
                void* dx, void* gout, size_t pixels, double count, hipStream_t st, const PoolSrc* pool = nullptr, int g_in_reduce = 0,
                const float* sum_rows = nullptr, int n_sum_rows = 0, int nseg = 1, int seg_stride = 0) {
  sslcr_ctx* c = n->ctx;
  BnBwdArgs a;
  if (nseg > 1) {                                // the passes as segments: nseg consecutive ring slots (the caller checked there are)
    double* sums = take_sums_n(c, nseg);
    if (!sums) return fail("bn_backward: no ring slots for the segments");
    TRYI(bn_bwd_begin(n, bn, sv, dy, x, yact, relu_from_x, dx, gout, pixels, count, st, sums, &a, pool, g_in_reduce, sum_rows, n_sum_rows, true,
                      nseg, seg_stride));
    return bn_bwd_end(c, a, st);
  }
  double* ring = sum_rows ? nullptr : take_sums(c);
  double* sums = ring ? ring : c->bn_sums;
  TRYI(bn_bwd_begin(n, bn, sv, dy, x, yact, relu_from_x, dx, gout, pixels, count, st, sums, &a, pool, g_in_reduce, sum_rows, n_sum_rows, ring != nullptr));
  TRYI(bn_bwd_sync(c, sums, 2 * (size_t)bn.C, st));
  return bn_bwd_end(c, a, st);
}

// the next writer of scratch buffer `kind` waits for the side-stream weight gradient that reads it
int wg_wait(sslcr_ctx* c, int kind, hipStream_t st) {
  if (c->wg_pending[kind]) {
    TRY(hipStreamWaitEvent(st, c->ev_wg_done[kind], 0));
    c->wg_pending[kind] = false;
  }
  return 0;
}
// `waiter` (the compute stream before the optimizer, the collective stream before a bucket) waits for every side-stream wgrad so far
int wg_join(sslcr_ctx* c, hipStream_t waiter) {
  if (!c->wg_any) return 0;
  TRY(hipEventRecord(c->ev_wg_join, c->wg_stream));
  TRY(hipStreamWaitEvent(waiter, c->ev_wg_join, 0));
  return 0;
}

// seg_images > 0: the N images are N / seg_images segments (TripletNet branches) whose producer BatchNorms sit seg_stride floats apart
// kind: which scratch buffer holds dy (0 conv2 / dRaw2, 1 conv1 / dRaw1, 2 projection / dRawD); < 0 = always on the compute stream
int wgrad_call(sslcr_net* n, const ConvL& L, const void* x, const void* dy, const BnSaved* pro, int N, int H, int W, int OH, int OW, hipStream_t st,
               int seg_images = 0, int seg_stride = 0, int kind = -1) {
  if (!n->rg[L.pidx]) return 0;
  sslcr_ctx* c = n->ctx;
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dy = dy; a.dw = (float*)n->grads.p + n->goff[L.pidx];
  if (pro) { a.in_scale = pro->scale; a.in_shift = pro->shift; a.in_relu = 1; }
  a.N = N; a.H = H; a.W = W; a.C = L.cin; a.K = L.cout; a.R = L.k; a.S = L.k; a.stride = L.stride; a.pad = L.pad; a.OH = OH; a.OW = OW;
  a.seg_images = seg_images; a.seg_stride = seg_stride;
  if (kind < 0 || !c->use_wg || c->prof.on) {
    TRY(prof_wgrad(c, c->dtype, a, st));
    return 0;
  }
  if (!c->wg_stream) {
    TRY(hipStreamCreateWithFlags(&c->wg_stream, hipStreamNonBlocking));
    for (int k = 0; k < 3; ++k) {
      TRY(hipEventCreateWithFlags(&c->ev_wg_in[k], hipEventDisableTiming));
      TRY(hipEventCreateWithFlags(&c->ev_wg_done[k], hipEventDisableTiming));
    }
    TRY(hipEventCreateWithFlags(&c->ev_wg_join, hipEventDisableTiming));
  }
  TRY(hipEventRecord(c->ev_wg_in[kind], st));                  // dy (and, on the first launch, the zeroed gradient buffer) is ready
  TRY(hipStreamWaitEvent(c->wg_stream, c->ev_wg_in[kind], 0));
  TRY(prof_wgrad(c, c->dtype, a, c->wg_stream));
  TRY(hipEventRecord(c->ev_wg_done[kind], c->wg_stream));
  c->wg_pending[kind] = true;
  c->wg_any = true;
  return 0;
}

int launch_bucket_allreduce(sslcr_net* n, int bucket, size_t lo, size_t hi, hipStream_t st) {
  sslcr_ctx* c = n->ctx;
  if (!sharded(c) || hi <= lo) return 0;
  TRY(hipEventRecord(c->ev_ready[bucket], st));
  TRY(hipStreamWaitEvent(c->comm_stream, c->ev_ready[bucket], 0));
  TRYI(wg_join(c, c->comm_stream));                             // ... and the weight gradients of this bucket launched on the side stream
  float* g = (float*)n->grads.p + lo;
  TRYI(all_reduce(c, 1, g, hi - lo, false, c->comm_stream));
  return 0;
}

// lowest parameter index that requires grad in the backbone (60 if none)
int lowest_trainable(const sslcr_net* n) {
  for (int i = 0; i < 60; ++i)
    if (n->rg[i]) return i;
  return 60;
}

// Backward of the backbone for all passes, walked LAYER-major: per block the BatchNorm backward / dgrad chain runs pass by
// pass (the reference's BatchNorm statistics are per branch), the weight gradients of conv1 and of the projection run ONCE
// over the branches' contiguous (x, dy) tensors (alloc_passes; scratch buffers of one kind are contiguous across passes too).
int backbone_backward(sslcr_net* n, int npass, hipStream_t st) {
  sslcr_ctx* c = n->ctx;
  const int dt = c->dtype;
  const size_t es = c->esz();
  PassState* P = n->pass;
  const int N = P[0].N;
  const Dims d = make_dims(P[0].H, P[0].W);
  const int low = lowest_trainable(n);
  if (low >= 60) return 0;
  // transient buffers, one layer1-sized unit each, laid out [kind][pass]: dOut, G, dRaw2, dAct1, dRaw1, dXin, dRawD; behind
  // them one stem-sized region for dRaw0 (used pass by pass)
  const size_t unit = (((size_t)N * d.ph * d.pw * 64 * es) + 255) & ~(size_t)255;
  const size_t stem_sz = (((size_t)N * d.oh0 * d.ow0 * 64 * es) + 255) & ~(size_t)255;
  TRYI(c->scratch.ensure(7 * npass * unit + stem_sz));
  char* S = (char*)c->scratch.p;
  // pass p of a kind sits p * (tensor bytes of the current layer) behind pass 0: contiguous for the batched weight gradients
  auto buf = [&](int kind, int p, size_t bytes) { return S + (size_t)kind * npass * unit + (size_t)p * bytes; };
  int kOut = 0, kXin = 5;
  const int kG = 1, kRaw2 = 2, kAct1 = 3, kRaw1 = 4, kRawD = 6;
  char* dRaw0 = S + (size_t)7 * npass * unit;
  // debug tap (pass 0 only): slot 0 G, 1 dRaw2, 2 dAct1, 3 dRaw1, 4 dRawD, 5 dXin
  auto tap = [&](int blk, int slot, const void* src, int n_, int h_, int w_, int c_) -> int {
    if (!n->tap) return 0;
    const size_t bytes = (size_t)n_ * h_ * w_ * c_ * es;
    TRYI(n->tapbuf[blk][slot].ensure(bytes));
    TRY(hipMemcpyAsync(n->tapbuf[blk][slot].p, src, bytes, hipMemcpyDeviceToDevice, st));
    int* d4 = n->tapdims[blk][slot];
    d4[0] = n_; d4[1] = h_; d4[2] = w_; d4[3] = c_;
    return 0;
  };

  size_t hi_pending = n->goff[60];
  int bucket = 1;
  for (int p = 0; p < npass; ++p)
    TRY(launch_avgpool_bwd(dt, n->dE[p], buf(kOut, p, (size_t)N * d.lh[7] * d.lw[7] * 512 * es), N, d.lh[7] * d.lw[7], 512, st));
  for (int i = 7; i >= 0; --i) {
    BlockL& B = n->blocks[i];
    const int oh = d.lh[i], ow = d.lw[i];
    const int xh = i == 0 ? d.ph : d.lh[i - 1], xw = i == 0 ? d.pw : d.lw[i - 1];
    const size_t opix = (size_t)N * oh * ow;
    const bool need_dx = low < B.pstart;          // something upstream of this block is trainable
    const size_t so = opix * B.c2.cout * es, si = (size_t)N * xh * xw * B.c1.cin * es;     // bytes of this block's output / input per pass
    // conv2's weight gradient applies bn1 + ReLU of its pass to the input on the fly: the halo kernel takes the passes as
    // segments with their own (scale, shift); other shapes go pass by pass
    bool c2_batched = false;
    if (npass > 1) {
      WgradArgs q;
      memset(&q, 0, sizeof(q));
      q.N = N * npass; q.H = oh; q.W = ow; q.C = B.c2.cin; q.K = B.c2.cout; q.R = 3; q.S = 3; q.stride = 1; q.pad = 1; q.OH = oh; q.OW = ow;
      c2_batched = wgrad_halo_tw(q) != 0 && (wgrad_halo_tw(q) == 16 || N % 2 == 0);
    }
    // ... and conv2's dgrad with the BatchNorm-backward front end takes them as segments where the 16x16-tile kernel serves it
    ConvArgs seg_m;
    bool seg_dg = false;
    if (npass > 1 && segments_on() && !c->prof.on) {
      seg_m = conv_args(B.c2, nullptr, B.c2.w_dg, nullptr, N * npass, oh, ow);
      seg_m.C = B.c2.cout; seg_m.K = B.c2.cin; seg_m.transposed = 0; seg_m.PH = oh; seg_m.PW = ow; seg_m.OH = oh; seg_m.OW = ow;
      const BnSaved& s1 = P[0].bn[B.b1.bidx];
      seg_m.mask_x = P[0].blk[i].raw1; seg_m.mask_scale = s1.scale; seg_m.mask_shift = s1.shift; seg_m.mask_mean = s1.mean;
      seg_m.seg_images = N; seg_m.seg_stride = (int)(P[1].bn[B.b1.bidx].scale - s1.scale);
      seg_dg = conv_h16_ok(dt, seg_m) && conv_segments_ok(dt, seg_m) && conv_partials_rows(seg_m) % npass == 0;
    }
    // the elementwise BatchNorm-backward passes of the branches as segments of one launch (sslcr_bn_bwd_desc.nseg): the passes'
    // saved tensors and scratch tensors are contiguous, their saved statistics seg_stride floats apart, their sums in
    // consecutive ring slots.  Not with synced BatchNorm (one all-reduce per pass) and not under the profiler.
    const bool seg_bn = npass > 1 && segments_on() && !c->prof.on && !(sharded(c) && c->bn_sync) && c->bn_ring.p &&
                        c->bn_ring_i + 2 * npass <= sslcr_ctx::kBnRing;
    const int bn_stride = npass > 1 ? (int)(P[1].bn[0].scale - P[0].bn[0].scale) : 0;
    if (seg_bn) {
      char *dOut = buf(kOut, 0, so), *G = buf(kG, 0, so), *dRaw2 = buf(kRaw2, 0, so), *dRawD = buf(kRawD, 0, so);
      TRYI(wg_wait(c, 0, st));
      if (B.has_ds) TRYI(wg_wait(c, 2, st));
      const size_t apix = opix * npass;
      if (B.has_ds) {
        BnBwdArgs a2, ad;
        double* sums_2 = take_sums_n(c, npass);
        if (!sums_2) return fail("backbone_backward: ring slots");
        TRYI(bn_bwd_begin(n, B.b2, P[0].bn[B.b2.bidx], dOut, P[0].blk[i].raw2, P[0].blk[i].y, 0, dRaw2, G, apix, (double)opix, st, sums_2, &a2, nullptr, 1,
                          nullptr, 0, true, npass, bn_stride));
        TRYI(bn_bwd_begin(n, B.bd, P[0].bn[B.bd.bidx], G, P[0].blk[i].rawd, nullptr, 0, dRawD, nullptr, apix, (double)opix, st, sums_2 + 2 * B.b2.C, &ad,
                          nullptr, 0, nullptr, 0, true, npass, bn_stride));
        TRYI(bn_bwd_end(c, a2, st));
        TRYI(bn_bwd_end(c, ad, st));
      } else {
        TRYI(bn_backward(n, B.b2, P[0].bn[B.b2.bidx], dOut, P[0].blk[i].raw2, P[0].blk[i].y, 0, dRaw2, G, apix, (double)opix, st, nullptr, 1, nullptr, 0,
                         npass, bn_stride));
      }
      TRYI(tap(i, 0, G, N, oh, ow, B.c2.cout));
      TRYI(tap(i, 1, dRaw2, N, oh, ow, B.c2.cout));
      if (B.has_ds) TRYI(tap(i, 4, dRawD, N, oh, ow, B.ds.cout));
    }
    for (int p = 0; p < npass; ++p) {
      PassState& ps = P[p];
      char *dOut = buf(kOut, p, so), *G = buf(kG, p, so), *dRaw2 = buf(kRaw2, p, so), *dAct1 = buf(kAct1, p, so), *dRaw1 = buf(kRaw1, p, so),
           *dRawD = buf(kRawD, p, so);
      // bn2 (+ relu mask from the block output): its REDUCE pass leaves G = dOut * (y > 0) in scratch; the apply pass, the
      // projection shortcut's BatchNorm and the identity path all read G instead of (dOut, y) again
      // (the previous block's side-stream weight gradients may still be reading dRaw2 / dRawD: order these writers behind them)
      if (!seg_bn) {
      TRYI(wg_wait(c, 0, st));
      if (B.has_ds) TRYI(wg_wait(c, 2, st));
      // ... and with a projection shortcut the two reduce passes run back to back, so that sharded runs exchange both
      // BatchNorms' sums ([2][C] each, adjacent in bn_sums) in ONE all-reduce before the two apply passes
      if (B.has_ds) {
        BnBwdArgs a2, ad;
        double* ring = take_sums(c);
        double* sums_2 = ring ? ring : c->bn_sums;
        double* sums_d = sums_2 + 2 * B.b2.C;
        // both BatchNorms receive the same g = dOut * (y > 0): one reduce pass forms it, writes it once and leaves both pairs of
        // sums, one apply pass reads it once for both (bn_bwd_reduce_pair_kernel); where that form does not apply, two passes each
        TRYI(bn_bwd_begin(n, B.b2, ps.bn[B.b2.bidx], dOut, ps.blk[i].raw2, ps.blk[i].y, 0, dRaw2, G, opix, (double)opix, st, sums_2, &a2, nullptr, 1,
                          nullptr, 0, ring != nullptr, 1, 0, true));
        TRYI(bn_bwd_begin(n, B.bd, ps.bn[B.bd.bidx], G, ps.blk[i].rawd, nullptr, 0, dRawD, nullptr, opix, (double)opix, st, sums_d, &ad, nullptr, 0,
                          nullptr, 0, ring != nullptr, 1, 0, true));
        const bool pair = bn_bwd_pair_ok(a2, ad);
        // (g is read by nobody else in a downsampling block, so with the mask as bits it need not be written at all -- measured
        //  SLOWER, 15.81 -> 15.90 ms same box: forming g from (dOut, bits) a second time costs the apply pass more than reading it.
        //  SSLCR_BN_PAIR_G=0 selects that form; the debug tap always keeps g)
        static const bool g_free = [] { const char* e = getenv("SSLCR_BN_PAIR_G"); return e && atoi(e) == 0; }();
        const int keep_g = (n->tap || !a2.yact_bits || !g_free) ? 1 : 0;
        if (pair) {
          TRY(launch_bn_bwd_reduce_pair(dt, a2, ad, keep_g, st));
        } else {
          TRY(launch_bn_bwd_reduce(dt, a2, st));
          TRY(launch_bn_bwd_reduce(dt, ad, st));
        }
        TRYI(bn_bwd_sync(c, sums_2, 4 * (size_t)B.b2.C, st));
        if (pair) {
          if (c->prof.on) {
            ProfRec r;
            r.e0 = c->prof.get(); r.e1 = c->prof.get();
            r.name = dt == DT_BF16 ? "sslcr::bn_bwd_apply_pair_kernel<unsigned short>" : "sslcr::bn_bwd_apply_pair_kernel<float>";
            r.flops = 0.0;
            r.bytes = (keep_g ? 5.0 : 5.0625) * (double)opix * B.b2.C * c->esz();   // reads g (or dy + mask bits), x of both BatchNorms; writes both dx
            (void)hipEventRecord(r.e0, st);
            hipError_t e = launch_bn_bwd_apply_pair(dt, a2, ad, keep_g, st);
            (void)hipEventRecord(r.e1, st);
            c->prof.rec[2].push_back(r);
            TRY(e);
          } else {
            TRY(launch_bn_bwd_apply_pair(dt, a2, ad, keep_g, st));
          }
        } else {
          TRYI(bn_bwd_end(c, a2, st));
          TRYI(bn_bwd_end(c, ad, st));
        }
      } else {
        TRYI(bn_backward(n, B.b2, ps.bn[B.b2.bidx], dOut, ps.blk[i].raw2, ps.blk[i].y, 0, dRaw2, G, opix, (double)opix, st, nullptr, 1));
      }
      if (p == 0) {
        TRYI(tap(i, 0, G, N, oh, ow, B.c2.cout));
        TRYI(tap(i, 1, dRaw2, N, oh, ow, B.c2.cout));
        if (B.has_ds) TRYI(tap(i, 4, dRawD, N, oh, ow, B.ds.cout));
      }
      }
      if (!c2_batched) TRYI(wgrad_call(n, B.c2, ps.blk[i].raw1, dRaw2, &ps.bn[B.b1.bidx], N, oh, ow, oh, ow, st, 0, 0, 0));
      if (seg_dg) continue;                      // conv2's dgrad runs once over the passes, below
      float* b1_rows = nullptr;
      int b1_nrows = 0;
      {
        ConvArgs a = conv_args(B.c2, dRaw2, B.c2.w_dg, dAct1, N, oh, ow);      // dgrad 3x3/1: gather over dRaw2 [.,K] with [C][R][S][K]
        a.C = B.c2.cout; a.K = B.c2.cin; a.transposed = 0; a.PH = oh; a.PW = ow; a.OH = oh; a.OW = ow;   // flipped pack
        // where the 16x16-tile kernel serves this dgrad it also applies bn1's ReLU mask and leaves bn1's two backward sums
        // in its stats rows: bn1's reduce pass over (dAct1, raw1) is not run
        ConvArgs m = a;
        const BnSaved& s1 = ps.bn[B.b1.bidx];
        m.mask_x = ps.blk[i].raw1; m.mask_scale = s1.scale; m.mask_shift = s1.shift; m.mask_mean = s1.mean;
        if (conv_h16_ok(dt, m)) {
          TRYI(ensure_partials(c, m, &b1_rows, &b1_nrows));
          m.stats = b1_rows;
          a = m;
        }
        TRY(prof_conv(c, dt,a, st));
        if (p == 0) {
          TRYI(tap(i, 2, dAct1, N, oh, ow, B.c2.cin));
          n->tap_flags[i] = b1_rows ? 1 : 0;
        }
      }
      TRYI(wg_wait(c, 1, st));
      TRYI(bn_backward(n, B.b1, ps.bn[B.b1.bidx], dAct1, ps.blk[i].raw1, nullptr, b1_rows ? 0 : 1, dRaw1, nullptr, opix, (double)opix, st,
                       nullptr, 0, b1_rows, b1_nrows));
      if (p == 0) TRYI(tap(i, 3, dRaw1, N, oh, ow, B.c1.cout));
    }
    if (seg_dg) {
      // conv2's dgrad of the three branches as segments of one launch (their scratch tensors are contiguous): segment s masks
      // with its own bn1 and leaves its own rows of bn1's backward sums
      float* rows = nullptr;
      int nrows = 0;
      seg_m.x = buf(kRaw2, 0, so); seg_m.y = buf(kAct1, 0, so);
      TRYI(ensure_partials(c, seg_m, &rows, &nrows));
      seg_m.stats = rows;
      TRY(prof_conv(c, dt, seg_m, st));
      TRYI(tap(i, 2, buf(kAct1, 0, so), N, oh, ow, B.c2.cin));
      n->tap_flags[i] = 1;
      const int per = nrows / npass;
      if (seg_bn) {
        TRYI(wg_wait(c, 1, st));
        TRYI(bn_backward(n, B.b1, P[0].bn[B.b1.bidx], buf(kAct1, 0, so), P[0].blk[i].raw1, nullptr, 0, buf(kRaw1, 0, so), nullptr, opix * npass,
                         (double)opix, st, nullptr, 0, rows, nrows, npass, bn_stride));
        TRYI(tap(i, 3, buf(kRaw1, 0, so), N, oh, ow, B.c1.cout));
      }
      for (int p = 0; p < npass && !seg_bn; ++p) {
        PassState& ps = P[p];
        TRYI(wg_wait(c, 1, st));
        TRYI(bn_backward(n, B.b1, ps.bn[B.b1.bidx], buf(kAct1, p, so), ps.blk[i].raw1, nullptr, 0, buf(kRaw1, p, so), nullptr, opix, (double)opix, st,
                         nullptr, 0, rows + (size_t)p * per * 2 * B.b1.C, per));
        if (p == 0) TRYI(tap(i, 3, buf(kRaw1, 0, so), N, oh, ow, B.c1.cout));
      }
    }
    // weight gradients: one launch over the npass * N images (x and dy contiguous across passes)
    if (c2_batched)
      TRYI(wgrad_call(n, B.c2, P[0].blk[i].raw1, buf(kRaw2, 0, so), &P[0].bn[B.b1.bidx], N * npass, oh, ow, oh, ow, st, N,
                      (int)(P[1].bn[B.b1.bidx].scale - P[0].bn[B.b1.bidx].scale), 0));
    {
      const char* X0 = i == 0 ? P[0].pooled : P[0].blk[i - 1].y;
      TRYI(wgrad_call(n, B.c1, X0, buf(kRaw1, 0, so), nullptr, N * npass, xh, xw, oh, ow, st, 0, 0, 1));
      if (B.has_ds) TRYI(wgrad_call(n, B.ds, X0, buf(kRawD, 0, so), nullptr, N * npass, xh, xw, oh, ow, st, 0, 0, 2));
    }
    if (need_dx) {
      {
        // conv1's dgrad (and the projection's) has no BatchNorm in it and is independent per image: one launch over all passes
        const int NB = N * npass;
        char *G = buf(kG, 0, so), *dRaw1 = buf(kRaw1, 0, so), *dXin = buf(kXin, 0, si), *dRawD = buf(kRawD, 0, so);
        ConvArgs a = conv_args(B.c1, dRaw1, B.c1.w_dg, dXin, NB, oh, ow);
        a.C = B.c1.cout; a.K = B.c1.cin; a.transposed = (B.c1.stride == 1) ? 0 : 1; a.PH = xh; a.PW = xw; a.OH = xh; a.OW = xw;
        if (!B.has_ds) a.residual = G;
        if (B.c1.stride == 1) {
          TRY(prof_conv(c, dt,a, st));
        } else {
          // stride-2 dgrad: each input-pixel parity class (ph%2, pw%2) only sees the taps with (p + pad - r) even --
          // 1 + 2 + 2 + 4 = 9 taps instead of 36 tap visits with three quarters zero-gathered; one launch with the class on
          // grid z where the DMA-gather kernel serves the shape, else four launches
          ConvArgs q4 = a;
          q4.pix_mul = 2; q4.PH = xh / 2; q4.PW = xw / 2; q4.par4 = 1;
          const bool one = xh % 2 == 0 && xw % 2 == 0 && conv_dma_bp(dt, q4) != 0 && conv_dma_bp(DT_BF16, q4) == conv_dma_bp(dt, q4) &&
                           conv_halo_tw(dt, q4) == 0;
          if (one) TRY(prof_conv(c, dt, q4, st));
          for (int par = 0; par < (one ? 0 : 4); ++par) {
            ConvArgs q = a;
            const int ph_ = par >> 1, pw_ = par & 1;
            q.pix_mul = 2; q.pix_off_h = ph_; q.pix_off_w = pw_;
            q.PH = (xh - ph_ + 1) / 2; q.PW = (xw - pw_ + 1) / 2;
            if (q.PH <= 0 || q.PW <= 0) continue;
            unsigned m = 0;
            for (int r = 0; r < 3; ++r)
              for (int s2 = 0; s2 < 3; ++s2)
                if (((ph_ + 1 - r) & 1) == 0 && ((pw_ + 1 - s2) & 1) == 0) m |= 1u << (r * 3 + s2);
            q.tap_mask = m;
            TRY(prof_conv(c, dt,q, st));
          }
        }
        if (B.has_ds) {       // 1x1/2 projection: scatter-accumulate into the even positions of dXin
          ConvArgs s = conv_args(B.ds, dRawD, B.ds.w_dg, dXin, NB, oh, ow);
          s.C = B.ds.cout; s.K = B.ds.cin; s.stride = 1; s.pad = 0; s.PH = oh; s.PW = ow; s.OH = xh; s.OW = xw; s.osh = B.ds.stride;
          s.accumulate = 1;
          TRY(prof_conv(c, dt,s, st));
        }
        TRYI(tap(i, 5, dXin, N, xh, xw, B.c1.cin));
      }
      const int t = kOut; kOut = kXin; kXin = t;      // ping-pong: this block's input gradient is the next dOut
    }
    if (i == 6 || i == 4 || i == 2) {   // layer4 / layer3 / layer2 gradients are final: reduce them under the rest of backward
      TRYI(launch_bucket_allreduce(n, bucket++, n->goff[B.pstart], hi_pending, st));
      hi_pending = n->goff[B.pstart];
    }
    if (!need_dx) break;
  }
  if (low < 3) {
    // stem: maxpool+relu backward -> bn0 backward -> conv1 wgrad (no dgrad: the input is data), pass by pass.
    // dOut (= dP, the pooled gradient) of pass p sits in its scratch unit, dRaw0 in the stem-sized region behind the units.  The
    // max-pool + ReLU backward is folded into both BatchNorm-backward passes (the un-pooled gradient is never written out).
    for (int p = 0; p < npass; ++p) {
      PassState& ps = P[p];
      PoolSrc pool{buf(kOut, p, (size_t)N * d.ph * d.pw * 64 * es), ps.argmax, d.oh0, d.ow0, d.ph, d.pw, ps.pooled};
      const size_t spix = (size_t)N * d.oh0 * d.ow0;
      StemWgradArgs w;
      memset(&w, 0, sizeof(w));
      w.x = ps.x; w.x2 = ps.x2; w.n_split = ps.n_split; w.dy = dRaw0; w.dw = (float*)n->grads.p + n->goff[0];
      w.N = N; w.H = ps.H; w.W = ps.W; w.OH = d.oh0; w.OW = d.ow0; w.in_f32 = ps.in_f32;
      if (n->rg[0] && c->fuse_stem_bwd) {
        // conv1 wgrad derives its dY tiles from the pooled gradient itself: the apply pass and its 2 x (N x 128 x 128 x 64)
        // round trip through HBM are gone (sslcr_stem_wgrad_pool)
        BnBwdArgs a;
        double* ring = take_sums(c);
        double* sums0 = ring ? ring : c->bn_sums;
        TRYI(bn_bwd_begin(n, n->bn0, ps.bn[0], nullptr, ps.raw0, nullptr, 1, nullptr, nullptr, spix, (double)spix, st, sums0, &a, &pool, 0, nullptr, 0,
                          ring != nullptr));
        TRYI(bn_bwd_sync(c, sums0, 2 * 64, st));
        w.dy = nullptr;
        if (c->prof.on) {
          ProfRec r;
          r.e0 = c->prof.get(); r.e1 = c->prof.get();
          r.name = dt == DT_BF16 ? "sslcr::stem_wgrad_kernel<unsigned short, pool>" : "sslcr::stem_wgrad_kernel<float, pool>";
          r.flops = 0.0;        // listed with the HBM-bound kernels (0.1 flop per byte)
          const double t = (double)spix * 64 * c->esz();
          r.bytes = 0.25 * t + 0.25 * (double)spix * 64 + t + (double)N * 3 * ps.H * ps.W * (ps.in_f32 ? 4 : 1);
          (void)hipEventRecord(r.e0, st);
          hipError_t e = launch_stem_wgrad_pool(dt, w, a, st);
          (void)hipEventRecord(r.e1, st);
          c->prof.rec[2].push_back(r);
          TRY(e);
        } else {
          TRY(launch_stem_wgrad_pool(dt, w, a, st));
        }
        continue;
      }
      TRYI(bn_backward(n, n->bn0, ps.bn[0], nullptr, ps.raw0, nullptr, 1, dRaw0, nullptr, spix, (double)spix, st, &pool));
      if (n->rg[0]) TRY(launch_stem_wgrad(dt, w, st));
    }
  }
  TRYI(launch_bucket_allreduce(n, bucket, 0, hi_pending, st));
  return 0;
}

int net_forward(sslcr_net* n, int train, const void* const* xs, int in_f32, int N, int H, int W, float* feats, float* logits, hipStream_t st) {
  if (!(train ? n->packed_train : n->packed_eval)) TRYI(sslcr_net_pack(n, train ? 1 : 2, st));   // shadow weights are stale
  const int npass_ = n->triplet ? 3 : 1;
  if (n->n8 > 0 && !n->f8_cal[train ? 0 : 1]) {
    // fp8 delayed scaling needs one forward to have seen the data: the FIRST forward of a net in a mode runs twice -- a
    // calibration pass at scale 1 (amax recorded, BatchNorm running statistics untouched, outputs overwritten below), then the
    // real one with the scales it produced
    n->f8_calib_pass = true;
    TRYI(alloc_heads(n, N));
    int rc = 0;
    for (int i = 0; i < npass_ && rc == 0 && train; ++i)
      rc = backbone_forward_train(n, n->pass[i], xs[i], in_f32, N, H, W, n->triplet ? 1 : 3, st);
    if (!train) rc = backbone_forward_eval(n, xs, npass_, in_f32, N, H, W, n->dE, st);
    n->f8_calib_pass = false;
    if (rc) return rc;                           // not calibrated: the next forward tries again instead of running at scale 1
    n->f8_cal[train ? 0 : 1] = true;
  }
  if (n->n8 > 0) TRY(launch_fp8_scale_update(n->f8slots, n->n8, st));      // fp8 delayed scaling: last forward's amax -> this one's scales
  if (train) n->packed_eval = false;               // running statistics are about to change
  const int npass = n->triplet ? 3 : 1;
  TRYI(alloc_heads(n, N));
  float* E[3] = {nullptr, nullptr, nullptr};
  const bool segs = train && npass == 3 && segments_servable(n, N, H, W);
  if (train) n->last_segments = segs;
  if (segs) {
    TRYI(backbone_forward_train_segments(n, xs, in_f32, N, H, W, st));
    for (int i = 0; i < npass; ++i) E[i] = n->pass[i].E;
  }
  for (int i = 0; i < npass && !segs && train; ++i) {
    TRYI(backbone_forward_train(n, n->pass[i], xs[i], in_f32, N, H, W, n->triplet ? 1 : 3, st));
    E[i] = n->pass[i].E;
  }
  if (!train) {
    for (int i = 0; i < npass; ++i) E[i] = n->dE[i];         // borrow the dE buffers (unused in eval) for the embeddings
    TRYI(backbone_forward_eval(n, xs, npass, in_f32, N, H, W, E, st));
  }
  TRYI(heads_forward(n, E, npass, N, st));
  if (feats) TRY(hipMemcpyAsync(feats, n->feats, (size_t)N * 768 * 4, hipMemcpyDeviceToDevice, st));
  if (logits) TRY(hipMemcpyAsync(logits, n->logits, (size_t)N * n->ncls * 4, hipMemcpyDeviceToDevice, st));
  n->last_N = N; n->last_npass = train ? npass : 0;
  return 0;
}

int net_backward(sslcr_net* n, const float* dlogits, hipStream_t st) {
  sslcr_ctx* c = n->ctx;
  if (n->last_npass == 0) return fail("sslcr_net_backward: no train-mode forward to differentiate");
  const int N = n->last_N, npass = n->last_npass;
  TRYI(n->grads.ensure(n->grad_count * sizeof(float)));
  TRY(hipMemsetAsync(n->grads.p, 0, n->grad_count * sizeof(float), st));
  const bool bb = lowest_trainable(n) < 60;
  if (bb) {
    TRYI(c->bn_ring.ensure((size_t)sslcr_ctx::kBnRing * sslcr_ctx::kBnSlot * sizeof(double)));
    c->bn_ring_i = 0;
  }
  TRYI(heads_backward(n, dlogits, npass, N, bb, st));
  TRYI(launch_bucket_allreduce(n, 0, n->goff[60], n->goff[n->nparams], st));
  if (bb) {
    TRYI(backbone_backward(n, npass, st));
  }
  if (sharded(c)) {
    TRY(hipEventRecord(c->ev_done, c->comm_stream));
    TRY(hipStreamWaitEvent(st, c->ev_done, 0));
  }
  TRYI(wg_join(c, st));                       // the optimizer (and the next step's scratch writers) come after the side-stream wgrads
  c->wg_any = false;
  for (bool& pnd : c->wg_pending) pnd = false;
  return 0;
}

}  // namespace

extern "C" {

int sslcr_create(sslcr_ctx** out, int device, int dtype) {
  if (!out || (dtype != SSLCR_F32 && dtype != SSLCR_BF16 && dtype != SSLCR_FP8)) return fail("sslcr_create: invalid argument");
  TRY(hipSetDevice(device));
  sslcr_ctx* c = new sslcr_ctx();
  c->device = device; c->dtype = dtype == SSLCR_FP8 ? SSLCR_BF16 : dtype; c->fp8 = dtype == SSLCR_FP8;
  if (const char* e = getenv("SSLCR_FUSE_STEM_BWD")) c->fuse_stem_bwd = atoi(e) != 0;
  // bn_stage: [kMaxSeg][32][2][C] (sslcr_bn_finalize_desc.stage with segments)
  if (c->small.ensure((size_t)kMaxSeg * 32 * 2 * 512 * sizeof(double) + 2 * 2 * 512 * sizeof(double) + 2 * 512 * sizeof(float) + 64 * sizeof(int)) != 0) { delete c; return -1; }
  c->bn_stage = (double*)c->small.p;
  c->bn_sums = c->bn_stage + (size_t)kMaxSeg * 32 * 2 * 512;        // two [2][C] slots (bn2 + projection BatchNorm of a block share an all-reduce)
  c->ones = (float*)(c->bn_sums + 2 * 2 * 512);
  c->zeros = c->ones + 512;
  c->bn_tickets = (int*)(c->zeros + 512);
  TRY(hipMemset(c->bn_tickets, 0, 64 * sizeof(int)));
  TRY(launch_fill(c->ones, 512, 1.f, nullptr));
  TRY(launch_fill(c->zeros, 512, 0.f, nullptr));
  TRY(hipStreamSynchronize(nullptr));
  g_live_ctx.fetch_add(1);
  *out = c;
  return 0;
}

int sslcr_destroy(sslcr_ctx* c) {
  if (!c) return 0;
  (void)hipDeviceSynchronize();
  if (c->wg_stream) {
    (void)hipStreamDestroy(c->wg_stream);
    for (int k = 0; k < 3; ++k) { (void)hipEventDestroy(c->ev_wg_in[k]); (void)hipEventDestroy(c->ev_wg_done[k]); }
    (void)hipEventDestroy(c->ev_wg_join);
  }
  if (c->aux_stream) {
    (void)hipStreamDestroy(c->aux_stream);
    (void)hipEventDestroy(c->ev_aux_begin);
    (void)hipEventDestroy(c->ev_aux_end);
  }
  if (c->comm || c->vcomm) {
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->comm_g) ncclCommDestroy(c->comm_g);
    (void)hipStreamDestroy(c->comm_stream);
    for (int i = 0; i < 8; ++i) (void)hipEventDestroy(c->ev_ready[i]);
    (void)hipEventDestroy(c->ev_done);
  }
  c->scratch.release(); c->partials.release(); c->small.release(); c->bn_ring.release();
  // the per-stream fold scratch of the weight-gradient / BatchNorm-backward launches is process-global: freed with the LAST live
  // context only -- another context's host thread (virtual ranks) may hold a slab pointer it has not launched with yet
  if (g_live_ctx.fetch_sub(1) == 1) stream_scratch_release();
  delete c;
  return 0;
}

int sslcr_profile(sslcr_ctx* c, int enable) {
  if (!c) return fail("sslcr_profile: null");
  c->prof.on = enable != 0;
  if (enable) {
    for (int w = 0; w < 3; ++w) {
      for (auto& r : c->prof.rec[w]) { c->prof.pool.push_back(r.e0); c->prof.pool.push_back(r.e1); }
      c->prof.rec[w].clear();
    }
  }
  return 0;
}

int sslcr_profile_dump(sslcr_ctx* c, char* buf, size_t n) {
  if (!c || !buf || n < 64) return fail("sslcr_profile_dump: invalid argument");
  TRY(hipDeviceSynchronize());
  struct Row { const char* name; double launches, ms, flops, bytes; };
  std::vector<Row> rows;
  for (int w = 0; w < 3; ++w)
    for (auto& r : c->prof.rec[w]) {
      float t = 0.f;
      TRY(hipEventElapsedTime(&t, r.e0, r.e1));
      Row* hit = nullptr;
      for (auto& q : rows)
        if (strcmp(q.name, r.name) == 0) { hit = &q; break; }
      if (!hit) { rows.push_back(Row{r.name, 0, 0, 0, 0}); hit = &rows.back(); }
      hit->launches += 1; hit->ms += t; hit->flops += r.flops; hit->bytes += r.bytes;
    }
  size_t off = 0;
  for (auto& q : rows) {
    int k = snprintf(buf + off, n - off, "%s|%.0f|%.6f|%.6e|%.6e\n", q.name, q.launches, q.ms, q.flops, q.bytes);
    if (k < 0 || (size_t)k >= n - off) return fail("sslcr_profile_dump: buffer too small");
    off += k;
  }
  buf[off] = 0;
  return 0;
}

int sslcr_profile_read(sslcr_ctx* c, int which, double* out4) {
  if (!c || !out4 || which < 0 || which > 1) return fail("sslcr_profile_read: invalid argument");
  TRY(hipDeviceSynchronize());
  double ms = 0.0, fl = 0.0, by = 0.0;
  for (auto& r : c->prof.rec[which]) {
    float t = 0.f;
    TRY(hipEventElapsedTime(&t, r.e0, r.e1));
    ms += t; fl += r.flops; by += r.bytes;
  }
  out4[0] = (double)c->prof.rec[which].size(); out4[1] = ms; out4[2] = fl; out4[3] = by;
  return 0;
}

int sslcr_comm_unique_id(void* id256) {
  if (!id256) return fail("sslcr_comm_unique_id: null");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  TRYN(ncclGetUniqueId((ncclUniqueId*)id256));
  TRYN(ncclGetUniqueId((ncclUniqueId*)((char*)id256 + 128)));
  return 0;
}

int sslcr_set_aux_stream(sslcr_ctx* c, int on) {
  if (!c) return fail("sslcr_set_aux_stream: null");
  c->use_aux = on ? 1 : 0;
  return 0;
}

int sslcr_set_wgrad_stream(sslcr_ctx* c, int on) {
  if (!c) return fail("sslcr_set_wgrad_stream: null");
  c->use_wg = on ? 1 : 0;
  return 0;
}

int sslcr_set_bn_sync(sslcr_ctx* c, int on) {
  if (!c) return fail("sslcr_set_bn_sync: null");
  c->bn_sync = on ? 1 : 0;
  return 0;
}

int sslcr_comm_init(sslcr_ctx* c, const void* id256, int rank, int world) {
  if (!c || !id256 || world < 1 || rank < 0 || rank >= world) return fail("sslcr_comm_init: invalid argument");
  // world == 1 normally needs no communicator; SSLCR_COMM_SELFTEST=1 creates the two 1-rank communicators anyway so that the
  // whole collective code path (RCCL calls, side stream, events) can be exercised on a single-GPU box (tests/test_engine_gpu.py)
  if (world == 1 && !getenv("SSLCR_COMM_SELFTEST")) { c->rank = 0; c->world = 1; return 0; }
  TRY(hipSetDevice(c->device));
  ncclUniqueId id, idg;
  memcpy(&id, id256, sizeof(id));
  memcpy(&idg, (const char*)id256 + 128, sizeof(idg));
  TRYN(ncclCommInitRank(&c->comm, world, id, rank));
  TRYN(ncclCommInitRank(&c->comm_g, world, idg, rank));
  TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  for (int i = 0; i < 8; ++i) TRY(hipEventCreateWithFlags(&c->ev_ready[i], hipEventDisableTiming));
  TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  c->rank = rank; c->world = world;
  return 0;
}

int sslcr_comm_info(sslcr_ctx* c, int* rank, int* world, int* transport) {
  if (!c) return fail("sslcr_comm_info: null");
  int r = c->rank, w = c->world;
  if (c->comm) {                 // what RCCL itself says, not what the caller passed
    TRYN(ncclCommUserRank(c->comm, &r));
    TRYN(ncclCommCount(c->comm, &w));
  }
  if (rank) *rank = r;
  if (world) *world = w;
  if (transport) *transport = c->comm ? 1 : c->vcomm ? 2 : 0;
  return 0;
}

int sslcr_comm_all_reduce_f32(sslcr_ctx* c, float* buf, size_t n, void* stream) {
  if (!c || (!buf && n)) return fail("sslcr_comm_all_reduce_f32: invalid argument");
  if (!sharded(c) || n == 0) return 0;
  return all_reduce(c, 0, buf, n, false, (hipStream_t)stream);
}

int sslcr_vcomm_create(sslcr_vcomm** out, int world) {
  if (!out || world < 1 || world > VW_MAX) return fail("sslcr_vcomm_create: world must be 1..%d", VW_MAX);
  sslcr_vcomm* v = new sslcr_vcomm();
  v->world = world;
  *out = v;
  return 0;
}

int sslcr_vcomm_destroy(sslcr_vcomm* v) {
  if (!v) return 0;
  (void)hipDeviceSynchronize();
  for (auto& ch : v->ch)
    for (int ph = 0; ph < 2; ++ph)
      for (int q = 0; q < VW_MAX; ++q) {
        if (ch.slot[ph][q]) (void)hipFree(ch.slot[ph][q]);
        if (ch.ev_in[ph][q]) { (void)hipEventDestroy(ch.ev_in[ph][q]); (void)hipEventDestroy(ch.ev_out[ph][q]); }
      }
  delete v;
  return 0;
}

int sslcr_comm_init_virtual(sslcr_ctx* c, sslcr_vcomm* v, int rank) {
  if (!c || !v || rank < 0 || rank >= v->world) return fail("sslcr_comm_init_virtual: invalid argument");
  if (c->comm || c->vcomm) return fail("sslcr_comm_init_virtual: this context already has a communicator");
  TRY(hipSetDevice(c->device));
  TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  for (int i = 0; i < 8; ++i) TRY(hipEventCreateWithFlags(&c->ev_ready[i], hipEventDisableTiming));
  TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  c->vcomm = v; c->rank = rank; c->world = v->world;
  return 0;
}

int sslcr_net_create(sslcr_ctx* c, const sslcr_net_desc* d, sslcr_net** out) {
  if (!c || !d || !out || !d->params) return fail("sslcr_net_create: null");
  const int want = 64 + (d->head_kind == 0 ? 2 : 4);
  if (d->nparams != want) return fail("sslcr_net_create: expected %d parameters (64 net + classifier), got %d", want, d->nparams);
  if (d->num_classes < 1 || d->num_classes > 64) return fail("sslcr_net_create: num_classes out of range");
  sslcr_net* n = new sslcr_net();
  n->ctx = c; n->nparams = d->nparams; n->head_kind = d->head_kind; n->ncls = d->num_classes; n->triplet = d->triplet;
  n->params.assign(d->params, d->params + d->nparams);
  n->rg.assign(d->nparams, 1);
  if (d->requires_grad) n->rg.assign(d->requires_grad, d->requires_grad + d->nparams);
  for (int i = 0; i < 20; ++i) {
    n->bn_rm[i] = d->bn_running_mean[i]; n->bn_rv[i] = d->bn_running_var[i];
    n->bn_nbt[i] = d->bn_num_batches_tracked ? d->bn_num_batches_tracked[i] : nullptr;
  }
  build_topology(n);
  if (alloc_shadow(n) != 0) { delete n; return -1; }
  *out = n;
  return 0;
}

int sslcr_net_destroy(sslcr_net* n) {
  if (!n) return 0;
  (void)hipDeviceSynchronize();
  n->shadow.release(); n->grads.release(); n->heads.release(); n->descs.release(); n->f8buf.release();
  for (int i = 0; i < 3; ++i) n->pass[i].mem.release();
  for (auto& row : n->tapbuf)
    for (DevBuf& b : row) b.release();
  delete n;
  return 0;
}

int sslcr_net_set_requires_grad(sslcr_net* n, const uint8_t* flags) {
  if (!n || !flags) return fail("sslcr_net_set_requires_grad: null");
  n->rg.assign(flags, flags + n->nparams);
  n->ndesc = 0;
  return 0;
}

int sslcr_net_pack(sslcr_net* n, int mode, void* stream) {
  if (!n || !(mode & 3)) return fail("sslcr_net_pack: invalid argument");
  hipStream_t st = (hipStream_t)stream;
  TRYI(pack_conv_layer(n, n->stem, n->bn0, mode, st));
  for (int i = 0; i < 8; ++i) {
    BlockL& B = n->blocks[i];
    TRYI(pack_conv_layer(n, B.c1, B.b1, mode, st));
    TRYI(pack_conv_layer(n, B.c2, B.b2, mode, st));
    if (B.has_ds) TRYI(pack_conv_layer(n, B.ds, B.bd, mode, st));
  }
  if (mode & 1) n->packed_train = true;
  if (mode & 2) n->packed_eval = true;
  // a full re-pack is what the host asks for after it changed parameters behind the engine's back (load_state_dict, --resume):
  // the fp8 activation scales were calibrated on the previous weights, so the next forward of each mode calibrates again
  if (mode == 3) n->f8_cal[0] = n->f8_cal[1] = false;
  return 0;
}

int sslcr_net_forward(sslcr_net* n, int train, const void* x1, const void* x2, const void* x3, int in_f32, int N, int H, int W,
                      float* feats, float* logits, void* stream) {
  if (!n || !x1 || N < 1 || H < 32 || W < 32) return fail("sslcr_net_forward: invalid argument");
  if (n->triplet && (!x2 || !x3)) return fail("sslcr_net_forward: TripletNet needs three inputs");
  const void* xs[3] = {x1, x2, x3};
  return net_forward(n, train, xs, in_f32, N, H, W, feats, logits, (hipStream_t)stream);
}

int sslcr_net_backward(sslcr_net* n, const float* dlogits, void* stream) {
  if (!n || !dlogits) return fail("sslcr_net_backward: null");
  return net_backward(n, dlogits, (hipStream_t)stream);
}

int sslcr_net_segments_used(const sslcr_net* n) { return (n && n->last_segments) ? 1 : 0; }
int sslcr_net_debug_tap(sslcr_net* n, int on) {
  if (!n) return fail("sslcr_net_debug_tap: null net");
  n->tap = on != 0;
  if (!on)
    for (auto& row : n->tapbuf)
      for (DevBuf& b : row) b.release();
  return 0;
}

int sslcr_net_debug_tensor(sslcr_net* n, int block, int kind, void* out, size_t out_bytes, int* dims4, int* flags, void* stream) {
  if (!n || block < 0 || block > 7 || kind < 0 || kind > 14 || !dims4) return fail("sslcr_net_debug_tensor: invalid argument");
  sslcr_ctx* c = n->ctx;
  hipStream_t st = (hipStream_t)stream;
  const BlockL& B = n->blocks[block];
  const PassState& ps = n->pass[0];
  if (n->last_npass == 0 || !ps.mem.p) return fail("sslcr_net_debug_tensor: no train-mode forward");
  const Dims d = make_dims(ps.H, ps.W);
  const int oh = d.lh[block], ow = d.lw[block];
  const int xh = block == 0 ? d.ph : d.lh[block - 1], xw = block == 0 ? d.pw : d.lw[block - 1];
  const void* src = nullptr;
  size_t esz = c->esz();
  int dm[4] = {ps.N, oh, ow, B.c2.cout};
  if (kind <= 5) {
    if (!n->tap || !n->tapbuf[block][kind].p) return fail("sslcr_net_debug_tensor: nothing tapped for block %d kind %d", block, kind);
    src = n->tapbuf[block][kind].p;
    for (int j = 0; j < 4; ++j) dm[j] = n->tapdims[block][kind][j];
  } else if (kind == 6) { src = ps.blk[block].raw1; dm[3] = B.c1.cout; }
  else if (kind == 7) { src = ps.blk[block].raw2; }
  else if (kind == 8) { src = ps.blk[block].rawd; if (!B.has_ds) return fail("sslcr_net_debug_tensor: block %d has no projection", block); }
  else if (kind == 9) { src = ps.blk[block].y; }
  else if (kind == 10) { src = block == 0 ? ps.pooled : ps.blk[block - 1].y; dm[1] = xh; dm[2] = xw; dm[3] = B.c1.cin; }
  else {                        // 11..14: bn1's saved scale, shift, mean, invstd (fp32 [C])
    const BnSaved& sv = ps.bn[B.b1.bidx];
    const float* v[4] = {sv.scale, sv.shift, sv.mean, sv.invstd};
    src = v[kind - 11]; esz = 4; dm[0] = B.b1.C; dm[1] = dm[2] = dm[3] = 1;
  }
  for (int j = 0; j < 4; ++j) dims4[j] = dm[j];
  if (flags) *flags = n->tap_flags[block];
  if (!out) return 0;
  const size_t bytes = (size_t)dm[0] * dm[1] * dm[2] * dm[3] * esz;
  if (out_bytes < bytes) return fail("sslcr_net_debug_tensor: buffer of %zu bytes, tensor has %zu", out_bytes, bytes);
  TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

int sslcr_net_grad(sslcr_net* n, int pidx, float* out, void* stream) {
  if (!n || !out || pidx < 0 || pidx >= n->nparams) return fail("sslcr_net_grad: invalid argument");
  if (!n->grads.p) return fail("sslcr_net_grad: no backward has run");
  hipStream_t st = (hipStream_t)stream;
  const float* g = (const float*)n->grads.p + n->goff[pidx];
  const ConvL* L = nullptr;
  for (int i = 0; i < 8 && !L; ++i) {
    BlockL& B = n->blocks[i];
    if (B.c1.pidx == pidx) L = &B.c1;
    else if (B.c2.pidx == pidx) L = &B.c2;
    else if (B.has_ds && B.ds.pidx == pidx) L = &B.ds;
  }
  if (L) {
    TRY(launch_unpack_grad(g, out, L->cout, L->cin, L->k * L->k, st));
  } else {
    TRY(hipMemcpyAsync(out, g, (size_t)n->psize[pidx] * 4, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

int sslcr_net_optimizer_step(sslcr_net* n, const sslcr_opt_desc* o, float* const* s1, float* const* s2, void* stream) {
  if (!n || !o || !s1) return fail("sslcr_net_optimizer_step: null");
  if (!n->grads.p) return fail("sslcr_net_optimizer_step: no gradients (run a backward first)");
  hipStream_t st = (hipStream_t)stream;
  // (re)build the device descriptor table when the state pointers or the trainable set changed
  bool rebuild = n->ndesc == 0 || (int)n->st1.size() != n->nparams;
  if (!rebuild)
    for (int i = 0; i < n->nparams; ++i)
      if (n->st1[i] != s1[i] || n->st2[i] != (s2 ? s2[i] : nullptr)) { rebuild = true; break; }
  if (rebuild) {
    n->host_descs.clear();
    n->max_n = 0;
    std::vector<int2> host_chunks;
    int packed_convs = 0;
    for (int i = 0; i < n->nparams; ++i) {
      if (!n->rg[i]) continue;
      if (!s1[i] || (o->kind == 0 && (!s2 || !s2[i]))) return fail("sslcr_net_optimizer_step: missing optimizer state for parameter %d", i);
      sslcr_tensor_desc t;
      memset(&t, 0, sizeof(t));
      t.p = n->params[i]; t.g = (float*)n->grads.p + n->goff[i]; t.s1 = s1[i]; t.s2 = s2 ? s2[i] : nullptr; t.n = n->psize[i];
      for (int b = 0; b < 8; ++b) {
        BlockL& B = n->blocks[b];
        const ConvL* L = B.c1.pidx == i ? &B.c1 : B.c2.pidx == i ? &B.c2 : (B.has_ds && B.ds.pidx == i) ? &B.ds : nullptr;
        if (L) {
          t.K = L->cout; t.C = L->cin; t.RS = L->k * L->k;
          // the update writes the train-mode shadow weights of the new value (what pack_conv_layer(mode 1) would produce)
          t.w_fwd = L->w_fwd; t.w_dgrad = L->w_dg; t.pack_dtype = n->ctx->dtype; t.dgrad_flip = (L->k == 3 && L->stride == 1);
          ++packed_convs;
        }
      }
      // work list: OPT_CHUNK elements per entry
      const int ti = (int)n->host_descs.size();
      if (t.K > 0 && t.RS == 9 && t.w_fwd && t.w_dgrad && t.K % 16 == 0 && t.C % 16 == 0) {
        for (int tile = 0; tile < (t.K / 16) * (t.C / 16); ++tile) host_chunks.push_back({ti, -(tile + 1)});   // LDS-transposed tiles
      } else {
        for (int e = 0; e < t.n; e += OPT_CHUNK) host_chunks.push_back({ti, e});
      }
      n->host_descs.push_back(t);
      if (t.n > n->max_n) n->max_n = t.n;
    }
    n->ndesc = (int)n->host_descs.size();
    if (n->ndesc == 0) return 0;
    int total_convs = 0;
    for (int b = 0; b < 8; ++b) total_convs += n->blocks[b].has_ds ? 3 : 2;
    n->opt_packs_all = packed_convs == total_convs;
    n->nchunks = (int)host_chunks.size();
    TRYI(n->chunks.ensure(host_chunks.size() * sizeof(int2)));
    TRY(hipMemcpyAsync(n->chunks.p, host_chunks.data(), host_chunks.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    TRYI(n->descs.ensure(n->ndesc * sizeof(sslcr_tensor_desc)));
    TRY(hipMemcpyAsync(n->descs.p, n->host_descs.data(), n->ndesc * sizeof(sslcr_tensor_desc), hipMemcpyHostToDevice, st));
    TRY(hipStreamSynchronize(st));       // host_descs may be rebuilt before the copy would otherwise land
    n->st1.assign(s1, s1 + n->nparams);
    if (s2) n->st2.assign(s2, s2 + n->nparams); else n->st2.assign(n->nparams, nullptr);
  }
  if (n->ndesc == 0) return 0;
  // sharded runs: every rank's loss is already scaled by 1/(global batch), so the all-reduced SUM is the exact gradient
  TRY(launch_optimizer_chunks((const sslcr_tensor_desc*)n->descs.p, n->chunks.p, n->nchunks, *o, st));
  n->packed_train = false; n->packed_eval = false;
  if (n->opt_packs_all) {
    // every block conv's shadow weights were rewritten with the update; only the stem's (its own K order) are left
    TRYI(pack_conv_layer(n, n->stem, n->bn0, 1, st));
    if (n->ctx->fp8) {                 // e4m3 train packs of the updated filters (the bf16 packs were written by the update itself)
      for (int b = 0; b < 8; ++b) {
        ConvL* Ls[2] = {&n->blocks[b].c1, &n->blocks[b].c2};
        for (ConvL* L : Ls) {
          if (!L->w8_fwd) continue;
          PackFp8Args f;
          memset(&f, 0, sizeof(f));
          f.w = n->params[L->pidx]; f.K = L->cout; f.C = L->cin; f.eps = 1e-5f; f.w8 = L->w8_fwd; f.w_dequant = L->dq_fwd;
          TRY(launch_pack_fp8(f, st));
        }
      }
    }
    n->packed_train = true;
  }
  return 0;
}

int sslcr_net_lookahead(sslcr_net* n, float* const* cached, float alpha, void* stream) {
  if (!n || !cached) return fail("sslcr_net_lookahead: null");
  for (int i = 0; i < n->nparams; ++i)
    if (cached[i]) TRY(launch_axpby(n->params[i], cached[i], n->psize[i], alpha, 1, (hipStream_t)stream));
  n->packed_train = false; n->packed_eval = false;
  return 0;
}

int sslcr_net_ema_from(sslcr_net* t, sslcr_net* s, float decay, void* stream) {
  if (!t || !s || t->nparams != s->nparams) return fail("sslcr_net_ema_from: mismatched nets");
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < t->nparams; ++i) TRY(launch_axpby(t->params[i], s->params[i], t->psize[i], decay, 0, st));
  // BN buffers are copied (deepcopy semantics for running statistics)
  TRY(hipMemcpyAsync(t->bn_rm[0], s->bn_rm[0], 64 * 4, hipMemcpyDeviceToDevice, st));
  TRY(hipMemcpyAsync(t->bn_rv[0], s->bn_rv[0], 64 * 4, hipMemcpyDeviceToDevice, st));
  if (t->bn_nbt[0] && s->bn_nbt[0]) TRY(hipMemcpyAsync(t->bn_nbt[0], s->bn_nbt[0], 8, hipMemcpyDeviceToDevice, st));
  for (int b = 0; b < 8; ++b) {
    BlockL& B = t->blocks[b];
    const BnL* bns[3] = {&B.b1, &B.b2, B.has_ds ? &B.bd : nullptr};
    for (const BnL* bn : bns) {
      if (!bn) continue;
      TRY(hipMemcpyAsync(t->bn_rm[bn->bidx], s->bn_rm[bn->bidx], bn->C * 4, hipMemcpyDeviceToDevice, st));
      TRY(hipMemcpyAsync(t->bn_rv[bn->bidx], s->bn_rv[bn->bidx], bn->C * 4, hipMemcpyDeviceToDevice, st));
      if (t->bn_nbt[bn->bidx] && s->bn_nbt[bn->bidx])
        TRY(hipMemcpyAsync(t->bn_nbt[bn->bidx], s->bn_nbt[bn->bidx], 8, hipMemcpyDeviceToDevice, st));
    }
  }
  t->packed_train = false; t->packed_eval = false;
  return 0;
}

int sslcr_step_ssl_cr(sslcr_net* te, sslcr_net* stn, const sslcr_ssl_cr_desc* d, void* stream) {
  if (!te || !stn || !d || !d->x_student || !d->x_teacher || !d->losses) return fail("sslcr_step_ssl_cr: null");
  if (d->nx < 1 || d->nu < 1) return fail("sslcr_step_ssl_cr: nx, nu must be positive");
  if (te->ncls != stn->ncls) return fail("sslcr_step_ssl_cr: teacher/student class count differ");
  hipStream_t st = (hipStream_t)stream;
  const int Ns = d->nx + d->nu;
  TRYI(alloc_heads(stn, Ns));
  // teacher: eval + no_grad (eval_BreastPathQ_SSL_CR.py:43-44,77-79) -- on the second stream when sslcr_set_aux_stream asked
  // for it (the per-kernel profiler records each launch on the stream it runs on)
  sslcr_ctx* c = stn->ctx;
  const bool aux = c->use_aux && te->ctx == c && !c->prof.on;     // (per-kernel profiling: serialised, like the weight-gradient stream)
  hipStream_t tst = st;
  if (aux) {
    if (!c->aux_stream) {
      TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
      TRY(hipEventCreateWithFlags(&c->ev_aux_begin, hipEventDisableTiming));
      TRY(hipEventCreateWithFlags(&c->ev_aux_end, hipEventDisableTiming));
    }
    TRY(hipEventRecord(c->ev_aux_begin, st));                 // inputs ready, last step's backward done with the scratch
    TRY(hipStreamWaitEvent(c->aux_stream, c->ev_aux_begin, 0));
    tst = c->aux_stream;
  }
  const void* xt[3] = {d->x_teacher, nullptr, nullptr};
  TRYI(net_forward(te, 0, xt, d->in_f32, d->nu, d->H, d->W, nullptr, stn->logits_t, tst));
  if (aux) TRY(hipEventRecord(c->ev_aux_end, c->aux_stream));
  // student: train mode on cat(x, u_s) (:82-84)
  const void* xs[3] = {d->x_student, nullptr, nullptr};
  if (d->x_student2 && stn->triplet) return fail("sslcr_step_ssl_cr: split student input needs a single-branch net");
  stn->split_x2 = d->x_student2; stn->split_n = d->nx;
  const int frc = net_forward(stn, 1, xs, d->in_f32, Ns, d->H, d->W, d->feats, d->logits, st);
  stn->split_x2 = nullptr; stn->split_n = 0;
  if (frc) return frc;
  if (aux) TRY(hipStreamWaitEvent(st, c->ev_aux_end, 0));
  LossArgs L;
  memset(&L, 0, sizeof(L));
  L.kind = d->kind; L.logits = stn->logits; L.logits_t = stn->logits_t; L.target_f = d->target_f; L.target_i = d->target_i;
  L.dlogits = d->backward ? stn->dlogits : nullptr; L.out = d->losses; L.nx = d->nx; L.nu = d->nu; L.C = stn->ncls; L.lambda_u = d->lambda_u;
  L.inv_nx_global = 1.f / (float)(d->nx_global > 0 ? d->nx_global : d->nx);
  L.inv_nu_global = 1.f / (float)(d->nu_global > 0 ? d->nu_global : d->nu);
  if (d->kind == 0 && !d->target_f) return fail("sslcr_step_ssl_cr: mse needs target_f");
  if (d->kind == 1 && !d->target_i) return fail("sslcr_step_ssl_cr: ce needs target_i");
  TRY(launch_loss(L, st));
  if (d->logits_t) TRY(hipMemcpyAsync(d->logits_t, stn->logits_t, (size_t)d->nu * stn->ncls * 4, hipMemcpyDeviceToDevice, st));
  if (d->backward) TRYI(net_backward(stn, stn->dlogits, st));
  return 0;
}

int sslcr_step_supervised(sslcr_net* n, const sslcr_sup_desc* d, void* stream) {
  if (!n || !d || !d->x1 || !d->losses) return fail("sslcr_step_supervised: null");
  if (d->kind != 2 && d->kind != 3) return fail("sslcr_step_supervised: kind must be 2 (ce) or 3 (mse)");
  if (n->triplet && (!d->x2 || !d->x3)) return fail("sslcr_step_supervised: TripletNet needs three inputs");
  hipStream_t st = (hipStream_t)stream;
  const void* xs[3] = {d->x1, d->x2, d->x3};
  TRYI(net_forward(n, d->train, xs, d->in_f32, d->n, d->H, d->W, d->feats, d->logits, st));
  LossArgs L;
  memset(&L, 0, sizeof(L));
  L.kind = d->kind; L.logits = n->logits; L.target_f = d->target_f; L.target_i = d->target_i;
  L.dlogits = (d->train && d->backward) ? n->dlogits : nullptr; L.out = d->losses; L.nx = d->n; L.nu = 0; L.C = n->ncls;
  L.inv_nx_global = 1.f / (float)(d->n_global > 0 ? d->n_global : d->n); L.inv_nu_global = 1.f;
  if (d->kind == 3 && !d->target_f) return fail("sslcr_step_supervised: mse needs target_f");
  if (d->kind == 2 && !d->target_i) return fail("sslcr_step_supervised: ce needs target_i");
  TRY(launch_loss(L, st));
  if (d->train && d->backward) TRYI(net_backward(n, n->dlogits, st));
  return 0;
}

}  // extern "C"
