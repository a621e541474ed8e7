// Host runtime behind the C ABI (include/eagle.h): weight store + BatchNorm folding, static launch schedule for
// HRNet-W48 (+head) and YOLOv8-{n,s,m,l,x} built once per handle, per-batch execution on HIP streams (optionally
// replayed as a hipGraph), record transfer and the RCCL gather.  No torch, no MIOpen/hipBLASLt: every kernel launched
// here is one of this library's own (conv.hip, elementwise.hip, detect.hip, geom.hip).
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <cstdarg>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <cstddef>
#include "common.h"

namespace eagle {

void fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Err{code, buf};
}

void ensure_max_dynamic_lds(const void* fn, int bytes)
{
    static std::mutex m;
    static std::map<std::pair<const void*, int>, int> done;      // (kernel, device) -> bytes already granted
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(m);
    auto it = done.find({fn, dev});
    if (it != done.end() && it->second >= bytes) return;
    HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[{fn, dev}] = bytes;
}

static thread_local std::string g_create_error;
// hipGraph capture against the rest of the process.  While ANY stream of the process is being captured, HIP refuses every operation that touches the legacy stream
// (hipMemcpy, hipMemset, a null-stream launch: hipErrorStreamCaptureImplicit) in every thread — also with a thread-local capture of non-blocking streams.  Small
// batches capture by default since round 5, and a second handle on another host thread (the concurrency tests; a multi-handle server) uploads weights or runs an
// operator on the null stream whenever it likes.  So a capture is made exclusive: every C-ABI entry holds this mutex shared for its duration (outermost call only),
// the capturing thread trades its shared hold for the exclusive one around Begin ... EndCapture + instantiate.
// Round 6: the exclusive hold is only ever TRIED, for a bounded time (CAPTURE_TRY_MS).  Another handle's call may hold the mutex shared for seconds (a clip
// call) or be blocked in a collective that waits for THIS thread's own gather (one thread per GPU in one process): waiting for it without a bound was a deadlock.
// A capture that cannot get its turn is skipped — the step runs as plain launches, the same kernels in the same order — and tried again a few steps later.
static std::shared_timed_mutex g_capture_mutex;
static std::mutex g_gate_m;
static std::condition_variable g_gate_cv;
static int g_capture_waiting = 0;                  // threads queued for the exclusive hold: new API calls let them pass first (glibc's rwlock prefers readers; with
                                                   // several handles making overlapping calls a capture would otherwise never get its turn), for a bounded time
static constexpr int CAPTURE_TRY_MS = 10, GATE_WAIT_MS = 20;
static thread_local std::shared_lock<std::shared_timed_mutex>* t_api_lock = nullptr;
struct ApiGuard {
    std::shared_lock<std::shared_timed_mutex> lk;
    bool outer;
    ApiGuard() : lk(g_capture_mutex, std::defer_lock), outer(t_api_lock == nullptr)
    {
        if (!outer) return;
        {
            std::unique_lock<std::mutex> g(g_gate_m);      // (a condition variable, not a spin: ADVICE r5)
            g_gate_cv.wait_for(g, std::chrono::milliseconds(GATE_WAIT_MS), [] { return g_capture_waiting == 0; });
        }
        lk.lock(); t_api_lock = &lk;
    }
    ~ApiGuard() { if (outer) t_api_lock = nullptr; }
};
struct CaptureExclusive {       // inside an API call: shared -> (try) exclusive -> shared again.  ok == false: no capture this time
    std::shared_lock<std::shared_timed_mutex>* al;
    std::unique_lock<std::shared_timed_mutex> ex;
    bool ok = false;
    CaptureExclusive() : al(t_api_lock), ex(g_capture_mutex, std::defer_lock)
    {
        if (al && al->owns_lock()) al->unlock();
        { std::lock_guard<std::mutex> g(g_gate_m); ++g_capture_waiting; }
        ok = ex.try_lock_for(std::chrono::milliseconds(CAPTURE_TRY_MS));
        { std::lock_guard<std::mutex> g(g_gate_m); --g_capture_waiting; }
        g_gate_cv.notify_all();
        if (!ok && al) al->lock();
    }
    ~CaptureExclusive() { if (ok) { ex.unlock(); if (al) al->lock(); } }
};
static int g_dbg_skip = 0;      // developer bisection (eagle_debug "skip"): 1 HRNet, 2 detector, 4 decode + NMS, 8 geometry kernel, 16 preprocess, 32 heat-map maxima, 64 fuse_sum / pool / upsample ops, 128 convolutions

struct HostTensor { std::vector<int64_t> shape; std::vector<float> data; };

// A few host threads that copy caller frames (pageable memory) into the pinned staging ring in parallel: one memcpy thread moves
// 5-8 GB/s, the per-frame path needs 2.76 MB x ~1800 frames/s = 5 GB/s on top of the PCIe transfer itself.
class CopyPool {
public:
    explicit CopyPool(int n) { for (int i = 0; i < n; ++i) th_.emplace_back([this] { work(); }); }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // fn(k) for k in [0, n), spread over the workers; returns when all are done
    void run(int n, const std::function<void(int)>& fn)
    {
        if (n <= 0) return;
        std::unique_lock<std::mutex> g(m_);
        fn_ = &fn; next_ = 0; n_ = n; left_ = n;
        cv_.notify_all();
        done_.wait(g, [this] { return left_ == 0; });
        fn_ = nullptr;
    }
private:
    void work()
    {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return quit_ || (fn_ && next_ < n_); });
            if (quit_) return;
            const int k = next_++;
            const std::function<void(int)>* f = fn_;
            g.unlock();
            (*f)(k);
            g.lock();
            if (--left_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, n_ = 0, left_ = 0;
    bool quit_ = false;
};

// ------------------------------------------------------------------------------------------------------------
struct Op {
    enum Kind { CONV, OTHER, FORK, JOIN } kind = OTHER;
    std::function<void(hipStream_t)> run;
    double flop = 0;
    double bytes = 0;        // OTHER ops: algorithmic HBM bytes of one launch (inputs read once + outputs written once)
    const char* tag = "";
    int stream = 0;          // 0: the network's own stream; 1..3: HRNet branch streams (concurrent branches between fuses)
    int nbranch = 0;         // FORK/JOIN: number of side streams involved
};

struct Net {                      // one launch schedule + the device memory it owns
    std::vector<Op> ops;
    std::vector<std::unique_ptr<std::string>> names;   // layer-shape labels of the convolutions (Op::tag points into them)
    std::vector<void*> owned;
    std::multimap<size_t, void*> free_list[4];   // one pool per stream: a buffer is recycled only by work ordered after its last use
    int pool = 0;                                // pool of the stream the builder is currently emitting for
    std::map<void*, size_t> sizes;
    size_t bytes = 0;
    ~Net() { for (void* p : owned) (void)hipFree(p); }
    void* get(size_t b)
    {
        b = (b + 255) & ~(size_t)255;
        auto it = free_list[pool].find(b);
        if (it != free_list[pool].end()) { void* p = it->second; free_list[pool].erase(it); return p; }
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, b));
        HIP_CHECK(hipMemset(p, 0, b));
        owned.push_back(p); sizes[p] = b; bytes += b;
        return p;
    }
    void put(void* p) { if (p) free_list[pool].insert({sizes.at(p), p}); }
    void* upload(const void* src, size_t b)
    {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(b, 16)));
        HIP_CHECK(hipMemcpy(p, src, b, hipMemcpyHostToDevice));
        owned.push_back(p); bytes += b;
        return p;
    }
};

}  // namespace eagle

using namespace eagle;

struct EagleHandle {
    EagleConfig cfg;
    std::string err;
    std::map<std::string, HostTensor> weights;
    bool finalized = false;
    int prec = 0, det_prec = 0;              // precision family of the key-point network / of the detector (EagleConfig::det_precision)
    hipStream_t s_main = nullptr, s_det = nullptr, s_post = nullptr, s_copy = nullptr;
    hipStream_t s_br[3] = {nullptr, nullptr, nullptr};          // HRNet branches 1..3 (branch 0 stays on s_main)
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    bool multi_stream = true;
    hipEvent_t ev_pre = nullptr, ev_det = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    // two-deep software pipeline: geometry + record D2H of batch i overlap the networks of batch i+1
    struct StepBuf {
        ArgmaxPart* parts = nullptr;
        EagleFrameResult* d_out = nullptr;
        EagleFrameResult* h_out = nullptr;   // pinned
        unsigned* d_sat = nullptr;           // [batch] saturated stores per frame (EAGLE_PREC_F32S; ConvArgs::sat).  Lies SAT_PAD bytes in front of d_out
        unsigned* h_sat = nullptr;           // (and of h_out): one memset and one device-to-host copy serve both
        uint8_t* d_frames = nullptr;         // staging copy of the batch (stable pointer for the captured graph)
        uint8_t* h_frames = nullptr;         // pinned ring slot for caller frames that live in pageable memory (allocated on first use)
        bool copy_pending = false;           // ev_copy has been recorded for this slot
        hipEvent_t ev_compute = nullptr, ev_done = nullptr, ev_copy = nullptr;
        // hipGraph instances of this slot's network phase, one per frame count (the source is always d_frames when graphs are on: a device-fed call is staged
        // into it).  Round 5 kept ONE instance keyed on (source pointer, frame count): a call whose last step is ragged, or a caller walking a resident clip,
        // re-captured (~80 ms, process-exclusive) on every call
        std::map<int, hipGraphExec_t> graphs;
        int capture_backoff = 0, capture_skip = 0;       // steps to run eagerly before the next capture attempt after one that could not get its turn
    } sb[2];
    bool graph_broken = false;                           // a capture failed half-way: this handle runs plain launches from now on
    int graph_captures = 0, graph_skipped = 0;           // EagleTimings::graph_captures / graph_skipped
    std::unique_ptr<Net> hr, yo, misc, reid;
    // K16 appearance embeddings (OSNet-x0.25; built when the "reid.*" tensors were loaded): REID_NB crops per pass
    TView reid_in; float* reid_feats = nullptr; EagleCrop* reid_crops = nullptr; EagleCrop* reid_crops_h = nullptr; float* reid_feats_h = nullptr;
    eagle::Tracker* tracker = nullptr;       // K14 state of the clip being tracked (eagle_track_*)
    std::unique_ptr<CopyPool> pool;          // host-side copy workers (eagle_process_frames from pageable memory)
    // step buffers
    TView kp_in, det_in, logits;
    LetterBox lb;
    int hm_chunks = 64;                      // heat-map partials per channel and frame (fused: output tiles of the head convolution)
    ArgmaxPart* cur_parts = nullptr;         // where the head convolution of the step being enqueued writes its partials (fused K5)
    unsigned* cur_sat = nullptr;             // where the f32s kernels of the step being enqueued count saturated stores per frame
    unsigned* clip_sat = nullptr;            // the same for the passes of a clip session (read by eagle_clip_fetch), pinned host copy behind it
    unsigned* clip_sat_h = nullptr;
    long long sat_events = 0; int sat_frames = 0;      // of the last eagle_process_* / eagle_clip_fetch call
    bool fused_argmax = false;
    DetScratch ds;
    DetLevel levels[3];
    PostParams pp;
    bool warmed = false;
    int call_steps = 0;                      // steps of the eagle_process_* call being executed (graph_on)
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> conv_ev;
    struct Span { int k; double bytes; hipEvent_t a, b; };
    std::vector<hipEvent_t> span_pool; size_t span_used = 0;
    std::vector<Span> spans;                     // non-convolution launches of the step being profiled
    std::vector<EagleKernelTime> ktab;           // accumulated per kernel name since eagle_set_profiling(1)
    std::vector<const Op*> conv_ops;             // the convolution behind each conv_ev pair of the step being profiled
    EagleTimings timings{};
    double conv_flop_step = 0; int n_conv = 0, n_launch = 0;
    // clip session of the optical-flow cadence (eagle_clip_*)
    struct Clip {
        bool open = false;
        ClipView cv;
        uint8_t* g[3] = {nullptr, nullptr, nullptr};
        EagleFrameResult* recs = nullptr;
        MemList* mem = nullptr;
        ChainState* st = nullptr;          // loop state
        ChainState* st_op = nullptr;       // scratch state of eagle_clip_flow
        ChainState* h_st = nullptr;        // pinned staging of eagle_clip_flow
        ChainState* h_zero = nullptr;      // pinned: the initial loop state
        int* h_tail = nullptr;             // pinned: {stalled, error} read-back, [2] = the constant -1
        MemList* h_mem = nullptr;          // pinned staging of eagle_clip_get/set_keypoints
        hipEvent_t ev_gray = nullptr, ev_det = nullptr, ev_kp = nullptr, ev_loop = nullptr;
        uint8_t* ecc_small = nullptr;      // K17: [n, ecc_h, ecc_w] 0.15-scale gray images (built on the first eagle_clip_motion_ecc call)
        int ecc_h = 0, ecc_w = 0;
        int2* ecc_pairs = nullptr; eagle::EccResult* ecc_out = nullptr;   // device, n entries
        int ecc_next = -1, ecc_tmpl = -2;  // eagle_clip_motion_ecc called range after range: the frame the last call stopped in front of, and the template it left
                                           // (a clip frame, or -1 = the carried template); a failed alignment keeps the OLD template across calls too
    } clip;
    // boxmot's ECC object lives as long as the tracker: the last template survives the clip (carried by eagle_clip_motion_ecc, reset by eagle_track_open)
    uint8_t* ecc_prev = nullptr; int ecc_prev_h = 0, ecc_prev_w = 0; bool ecc_has_prev = false;
    // comm
    void* rccl = nullptr; void* comm = nullptr; int rank = 0, world = 1;
    void* gather_buf = nullptr; size_t gather_cap = 0;      // device staging of eagle_gather (send | receive), grown on demand
};

namespace eagle {

// ------------------------------------------------------------------------------------------------------------
// Builder: shared by both networks
// ------------------------------------------------------------------------------------------------------------
struct Builder {
    EagleHandle* H;
    Net* net;
    int prec;
    double bn_eps;
    int N;
    const char* label_suffix = "";          // appended to the convolution labels of this network (" d": the detector, so that profiles can tell the networks apart)
    int cur_stream = 0;
    void set_stream(int k) { cur_stream = k; net->pool = k; }
    void fork_join(Op::Kind kind, int nbranch) { Op op; op.kind = kind; op.nbranch = nbranch; op.tag = kind == Op::FORK ? "fork" : "join"; net->ops.push_back(op); }

    int gran() const { return prec == EAGLE_PREC_F32 ? 4 : 8; }

    TView act(int h, int w, int c, bool f32 = false)
    {
        TView v; v.n = N; v.h = h; v.w = w; v.c = c; v.cs = c; v.off = 0;
        v.f32 = (f32 || prec == EAGLE_PREC_F32) ? 1 : prec_tensor_fmt(prec);
        v.p = net->get((size_t)N * h * w * c * v.esize());
        return v;
    }
    void release(const TView& v) { net->put(v.p); }

    const HostTensor& W(const std::string& name)
    {
        auto it = H->weights.find(name);
        if (it == H->weights.end()) fail(EAGLE_E_MISSING, "weight tensor '%s' was not loaded", name.c_str());
        return it->second;
    }

    // conv (+ folded BN when bn != ""), output into `out` if given (a slice of a concat buffer), else a new tensor.
    TView conv(const TView& x, const std::string& cname, const std::string& bn, int stride, int pre, const TView* r1,
               const TView* r2, int post, const TView* out = nullptr, bool out_f32 = false, ArgmaxPart* const* am_slot = nullptr)
    {
        const HostTensor& w = W(cname + ".weight");
        if (w.shape.size() != 4) fail(EAGLE_E_INVALID, "%s.weight: expected 4-d", cname.c_str());
        const int cout = (int)w.shape[0], cin = (int)w.shape[1], ks = (int)w.shape[2];
        if (cin > x.c) fail(EAGLE_E_INVALID, "%s: input has %d channels, weight expects %d", cname.c_str(), x.c, cin);
        const int cout_pad = (cout + 15) / 16 * 16;
        // fold: scale = gamma / sqrt(var + eps) (float64); w' = f32(f64(w)*scale); b' = f32(beta - mean*scale)
        std::vector<float> hwio((size_t)ks * ks * cin * cout), bias(cout_pad, 0.f);
        std::vector<double> scale(cout, 1.0);
        if (!bn.empty()) {
            const HostTensor &g = W(bn + ".weight"), &b = W(bn + ".bias"), &m = W(bn + ".running_mean"), &v = W(bn + ".running_var");
            for (int o = 0; o < cout; ++o) {
                scale[o] = (double)g.data[o] / std::sqrt((double)v.data[o] + bn_eps);
                bias[o] = (float)((double)b.data[o] - (double)m.data[o] * scale[o]);
            }
        } else if (H->weights.count(cname + ".bias")) {
            const HostTensor& b = W(cname + ".bias");
            for (int o = 0; o < cout; ++o) bias[o] = b.data[o];
        }                                                       // (a linear convolution without BatchNorm and bias: OSNet's LightConv3x3.conv1)
        for (int o = 0; o < cout; ++o)
            for (int i = 0; i < cin; ++i)
                for (int t = 0; t < ks * ks; ++t) {
                    const float wv = w.data[((size_t)o * cin + i) * ks * ks + t];
                    hwio[((size_t)t * cin + i) * cout + o] = bn.empty() ? wv : (float)((double)wv * scale[o]);
                }
        const int ho = (x.h + 2 * (ks / 2) - ks) / stride + 1, wo = (x.w + 2 * (ks / 2) - ks) / stride + 1;
        if (const char* dump = getenv("EAGLE_DUMP_LAYERS")) {
            if (FILE* f = fopen(dump, "a")) { fprintf(f, "%d,%d,%d,%d,%d,%d,%d\n", ks, stride, x.c, cout_pad, x.h, x.w, N); fclose(f); }
        }
        ConvLaunch L;
        L.cfg = conv_choose(prec, ks, stride, x.c, cout_pad, wo, pre == ACT_NONE && post <= ACT_RELU && !out_f32, r1 && r2, r1 || r2);
        if (!conv_supported(prec, L.cfg))
            fail(EAGLE_E_NOKERNEL, "%s: no kernel instance (ks=%d s=%d kc=%d nt=%d)", cname.c_str(), ks, stride, L.cfg.kc, L.cfg.nt);
        const size_t ne = conv_weight_elems(prec, L.cfg);
        std::vector<char> tiled(ne * (prec == EAGLE_PREC_F32 ? 4 : 2));
        conv_tile_weights(prec, L.cfg, hwio.data(), cin, cout, tiled.data(), &L.descale);
        L.w = net->upload(tiled.data(), tiled.size());
        L.bias = (const float*)net->upload(bias.data(), bias.size() * 4);
        L.x = x;
        if (out) {
            L.y = *out;
            if (out->c != cout_pad || out->h != ho || out->w != wo) fail(EAGLE_E_INVALID, "%s: output slice mismatch", cname.c_str());
        } else {
            L.y = act(ho, wo, cout_pad, out_f32);
        }
        if (r1) L.r1 = *r1;
        if (r2) L.r2 = *r2;
        L.pre_act = pre; L.post_act = post; L.out_f32 = out_f32 ? 1 : 0;
        // every convolution kernel addresses its tensors through raw buffer descriptors with 32-bit byte offsets: a tensor that reaches 2 GiB is
        // refused HERE (eagle_finalize_weights returns the error), not at the first launch in the middle of a pipeline or a hipGraph capture
        for (const TView* t : {&L.x, &L.y, r1 ? &L.r1 : nullptr, r2 ? &L.r2 : nullptr})
            if (t && (size_t)t->n * t->h * t->w * t->cs * t->esize() >= ((size_t)1 << 31))
                fail(EAGLE_E_INVALID, "%s: a %d x %d x %d x %d-channel tensor of this layer reaches 2 GiB at a device batch of %d frames (32-bit tensor offsets); use a smaller EagleConfig.batch",
                     cname.c_str(), t->n, t->h, t->w, t->cs, N);
        L.am_slot = am_slot;
        if (prec == EAGLE_PREC_F32S && !out_f32 && !am_slot) L.sat_slot = &H->cur_sat;
        if (am_slot) { H->hm_chunks = conv_tiles_per_frame(L.cfg, ho, wo); H->fused_argmax = true; }
        L.flop = 2.0 * N * ho * wo * (double)cout * cin * ks * ks;
        const int pr = prec;
        char label[64];
        snprintf(label, sizeof(label), "conv %dx%d/%d %d->%d @%dx%d v%d%s", ks, ks, stride, cin, cout, ho, wo, L.cfg.variant, label_suffix);      // v: kernel form (conv.hip)
        net->names.emplace_back(new std::string(label));
        Op op; op.kind = Op::CONV; op.flop = L.flop; op.tag = net->names.back()->c_str(); op.stream = cur_stream;
        {   // algorithmic HBM bytes of the launch: input once, output once, each residual once, weights once
            const double es = prec == EAGLE_PREC_F16 ? 2 : 4;
            op.bytes = (double)N * x.h * x.w * cin * es + (double)N * ho * wo * cout * ((out_f32 || prec != EAGLE_PREC_F16) ? 4 : 2) * (am_slot ? 0 : 1) +
                       (r1 ? (double)N * ho * wo * cout * es : 0) + (r2 ? (double)N * ho * wo * cout * es : 0) + (double)ks * ks * cin * cout * es;
        }
        op.run = [L, pr](hipStream_t s) { conv_launch(pr, L, s); };
        net->ops.push_back(op);
        return L.y;
    }
    // folded weights of conv `cname` (+ BatchNorm `bn`) as [tap][cin][cout] and the folded bias: the arithmetic of conv() above, for the fused launches
    void fold(const std::string& cname, const std::string& bn, std::vector<float>& hwio, std::vector<float>& bias, int& cin, int& cout, int& ks)
    {
        const HostTensor& w = W(cname + ".weight");
        if (w.shape.size() != 4) fail(EAGLE_E_INVALID, "%s.weight: expected 4-d", cname.c_str());
        cout = (int)w.shape[0]; cin = (int)w.shape[1]; ks = (int)w.shape[2];
        hwio.assign((size_t)ks * ks * cin * cout, 0.f); bias.assign(cout, 0.f);
        std::vector<double> scale(cout, 1.0);
        const HostTensor &g = W(bn + ".weight"), &b = W(bn + ".bias"), &m = W(bn + ".running_mean"), &v = W(bn + ".running_var");
        for (int o = 0; o < cout; ++o) {
            scale[o] = (double)g.data[o] / std::sqrt((double)v.data[o] + bn_eps);
            bias[o] = (float)((double)b.data[o] - (double)m.data[o] * scale[o]);
        }
        for (int o = 0; o < cout; ++o)
            for (int i = 0; i < cin; ++i)
                for (int t = 0; t < ks * ks; ++t)
                    hwio[((size_t)t * cin + i) * cout + o] = (float)((double)w.data[((size_t)o * cin + i) * ks * ks + t] * scale[o]);
    }
    // One launch for a whole Bottleneck (bneck.hip): relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + res), kh.py:101-137
    // ds_conv / ds_bn non-empty (block 0: Cin = 64): the 1 x 1 downsample branch of the shortcut is computed inside the launch (conv3 over K = 128 = [t2 | x], one weight
    // scale, bias b3 + bd) and `res` is not read
    TView bottleneck(const TView& x, const std::string& q, const TView& res, const std::string& ds_conv = "", const std::string& ds_bn = "")
    {
        BneckLaunch L;
        const bool dsf = !ds_conv.empty();
        std::vector<float> hw, bs; int ci, co, ks;
        std::vector<_Float16> img;
        double flop = 0;
        const char* cn[3] = {"conv1", "conv2", "conv3"}; const char* bnn[3] = {"bn1", "bn2", "bn3"};
        const int want_ks[3] = {1, 3, 1}, want_ci[3] = {x.c, 64, 64}, want_co[3] = {64, 64, 256};
        for (int k = 0; k < 3; ++k) {
            fold(q + cn[k], q + bnn[k], hw, bs, ci, co, ks);
            if (ks != want_ks[k] || ci != want_ci[k] || co != want_co[k]) fail(EAGLE_E_INVALID, "%s%s: not the Bottleneck shape of the fused kernel", q.c_str(), cn[k]);
            if (k == 2 && dsf) {                            // conv3's K dimension extended by the downsample branch: rows 0 .. 63 = W3 (over t2), rows 64 .. 127 = Wd (over x)
                std::vector<float> hwd, bd; int cid, cod, ksd;
                fold(ds_conv, ds_bn, hwd, bd, cid, cod, ksd);
                if (ksd != 1 || cid != 64 || cod != 256 || x.c != 64) fail(EAGLE_E_INVALID, "%s: not the 1x1 64->256 downsample of the fused kernel", ds_conv.c_str());
                hw.insert(hw.end(), hwd.begin(), hwd.end());
                for (int o = 0; o < 256; ++o) bs[o] = bs[o] + bd[o];
                ci = 128;
                flop += 2.0 * N * x.h * x.w * 256.0 * 64;
            }
            float ds = 1.f;
            bneck_tile_weights(hw.data(), ks * ks, ci, co, img, &ds);
            bneck_scale_bias(bs, ds);
            const void* dw = net->upload(img.data(), img.size() * 2);
            const float* db = (const float*)net->upload(bs.data(), bs.size() * 4);
            if (k == 0) { L.w1 = dw; L.b1 = db; L.ds1 = ds; } else if (k == 1) { L.w2 = dw; L.b2 = db; L.ds2 = ds; } else { L.w3 = dw; L.b3 = db; L.ds3 = ds; }
            flop += 2.0 * N * x.h * x.w * (double)co * (k == 2 ? 64 : ci) * ks * ks;
        }
        L.x = x; L.y = act(x.h, x.w, 256); L.res = dsf ? L.y : res; L.ds_fused = dsf;
        L.sat_slot = &H->cur_sat;
        for (const TView* t : {&L.x, &L.y, &L.res})
            if ((size_t)t->n * t->h * t->w * t->cs * t->esize() >= ((size_t)1 << 31))
                fail(EAGLE_E_INVALID, "%s: a %d x %d x %d x %d-channel tensor of this block reaches 2 GiB at a device batch of %d frames (32-bit tensor offsets); use a smaller EagleConfig.batch",
                     q.c_str(), t->n, t->h, t->w, t->cs, N);
        char label[64];
        snprintf(label, sizeof(label), "bneck %d->64->256%s @%dx%d%s", x.c, dsf ? "+ds" : "", x.h, x.w, label_suffix);
        net->names.emplace_back(new std::string(label));
        Op op; op.kind = Op::CONV; op.flop = flop; op.tag = net->names.back()->c_str(); op.stream = cur_stream;
        op.bytes = (double)N * x.h * x.w * 4.0 * (x.c + 256 + ((dsf || res.p == x.p) ? 0 : 256)) + 4.0 * (x.c * 64 + 9 * 64 * 64 + 64 * 256 + (dsf ? 64 * 256 : 0));      // x once, y once (+ a separate residual tensor), weights once
        op.run = [L](hipStream_t s) { bneck_launch(L, s); };
        net->ops.push_back(op);
        return L.y;
    }
    static double vbytes(const TView& v) { return (double)v.n * v.h * v.w * v.c * (v.f32 ? 4 : 2); }
    void other(std::function<void(hipStream_t)> fn, const char* tag, double bytes = 0)
    {
        Op op; op.kind = Op::OTHER; op.run = std::move(fn); op.tag = tag; op.stream = cur_stream; op.bytes = bytes;
        net->ops.push_back(op);
    }
};

// ------------------------------------------------------------------------------------------------------------
// HRNet-W48 + head (eagle/models/keypoint_hrnet.py:315-351, 444-481, 283-309, 553-562).  The fusion plan
// (which adds ride in which epilogue) is the one oracle/nets.py::_hr_stage mirrors.
// ------------------------------------------------------------------------------------------------------------
static const char* HRP = "unnormalized_model.0.";

static std::vector<TView> hr_stage(Builder& B, std::vector<TView> xs, int stage_idx, int n_modules, int nb, bool last_single)
{
    const int R = ACT_RELU;
    for (int m = 0; m < n_modules; ++m) {
        const std::string q = std::string(HRP) + "stage" + std::to_string(stage_idx) + "." + std::to_string(m) + ".";
        B.fork_join(Op::FORK, nb - 1);          // the branches of a module are independent until the fuse
        // emission order (= host launch order, and node order of a captured graph): block k of every branch before block k + 1 of any, the widest branch first —
        // a branch's first launch is then not queued behind the whole chains of the branches before it: one-frame call 6.74 -> 6.63 ms, four frames 8.62 -> 8.46
        // (profiles/r06ag_*; EAGLE_HR_INTERLEAVE=0: branch after branch)
        const bool interleave = !(getenv("EAGLE_HR_INTERLEAVE") && atoi(getenv("EAGLE_HR_INTERLEAVE")) == 0);
        for (int step = 0; step < 4 * nb; ++step) {
            const int k = interleave ? step / nb : step % 4, b = interleave ? nb - 1 - step % nb : step / 4;
            B.set_stream(b);
            TView x = xs[b];
            const std::string r = q + "branches." + std::to_string(b) + "." + std::to_string(k) + ".";
            TView o = B.conv(x, r + "conv1", r + "bn1", 1, 0, nullptr, nullptr, R);
            TView y = B.conv(o, r + "conv2", r + "bn2", 1, 0, &x, nullptr, R);
            B.release(o);
            B.release(x);
            xs[b] = y;
        }
        B.set_stream(0);
        B.fork_join(Op::JOIN, nb - 1);
        const int n_out = (last_single && m == n_modules - 1) ? 1 : nb;
        std::vector<TView> out;
        // (round 6) The fuse outputs are independent of each other: output i — its chain of stride-2 convolutions from the higher-resolution branches, its 1 x 1
        // convolutions of the lower ones, its fuse_sum — is emitted for stream i, so that with branch streams on (every batch since this round) the module's 19 fuse launches are
        // four parallel chains of at most 6 instead of one serial chain behind the join: 7.5 -> 6.66 ms per one-frame call, +1.6 .. 2.8 % at B = 50 (r06aj, r06ak).  The inputs
        // xs[] are only read; everything a chain allocates and releases stays in its own stream's pool.  Same launches, same arithmetic.  (Measured and not kept: also
        // moving the launches that need only ONE branch's output in front of the join, behind that branch — nothing, twice: profiles/r06ai_*.)
        if (n_out > 1) B.fork_join(Op::FORK, n_out - 1);
        for (int i = 0; i < n_out; ++i) {
            B.set_stream(n_out > 1 ? i : 0);
            TView y; bool have_y = false;
            for (int j = 0; j < i; ++j) {
                TView t = xs[j];
                for (int k = 0; k < i - j; ++k) {
                    const std::string r = q + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + "." + std::to_string(k) + ".";
                    const bool last = k == i - j - 1;
                    TView nt;
                    if (!last) {
                        nt = B.conv(t, r + "0", r + "1", 2, 0, nullptr, nullptr, R);
                    } else {
                        const TView* ident = (j == i - 1) ? &xs[i] : nullptr;
                        const bool relu_now = ident && i == nb - 1;
                        nt = B.conv(t, r + "0", r + "1", 2, 0, have_y ? &y : nullptr, ident, relu_now ? R : 0);
                    }
                    if (k > 0) B.release(t);
                    t = nt;
                }
                if (have_y) B.release(y);
                y = t; have_y = true;
            }
            if (i == 0) y = xs[0];
            FuseUp ups[3]; int nu = 0;
            for (int j = i + 1; j < nb; ++j) {
                const std::string r = q + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + ".";
                ups[nu++].z = B.conv(xs[j], r + "0", r + "1", 1, 0, nullptr, nullptr, 0);
            }
            if (nu) {
                TView o = B.act(y.h, y.w, y.c);
                const TView base = y; FuseUp u0 = ups[0], u1 = ups[1], u2 = ups[2]; const int n_up = nu;
                EagleHandle* const Hh = B.H;
                B.other([base, u0, u1, u2, n_up, o, Hh](hipStream_t s) { FuseUp u[3] = {u0, u1, u2}; fuse_sum_launch(base, u, n_up, 1, o, s, Hh->cur_sat); }, "fuse_sum",
                        Builder::vbytes(base) + Builder::vbytes(o) + (nu > 0 ? Builder::vbytes(ups[0].z) : 0) + (nu > 1 ? Builder::vbytes(ups[1].z) : 0) + (nu > 2 ? Builder::vbytes(ups[2].z) : 0));
                for (int k = 0; k < nu; ++k) B.release(ups[k].z);
                if (i != 0) B.release(y);
                y = o;
            }
            out.push_back(y);
        }
        B.set_stream(0);
        if (n_out > 1) B.fork_join(Op::JOIN, n_out - 1);
        // inputs of this module's fuse are dead now (xs[0] may be aliased by out[0] only when nb == 1, never here)
        for (int b = 0; b < nb; ++b) B.release(xs[b]);
        xs = out;
    }
    return xs;
}

static TView build_hrnet(Builder& B, const TView& x_in)
{
    const int R = ACT_RELU;
    const std::string P = HRP;
    TView x = B.conv(x_in, P + "conv1", P + "bn1", 2, 0, nullptr, nullptr, R);
    TView x2 = B.conv(x, P + "conv2", P + "bn2", 2, 0, nullptr, nullptr, R);
    B.release(x); x = x2;
    for (int b = 0; b < 4; ++b) {
        const std::string q = P + "layer1." + std::to_string(b) + ".";
        // round 6: the whole Bottleneck as one launch in the split family (EAGLE_BNECK_FUSED=0: the three launches of rounds 1-5); block 0's downsample branch inside it
        // (EAGLE_BNECK_DS=0: as its own launch, the residual read back)
        const char* fe = getenv("EAGLE_BNECK_FUSED");
        const char* de = getenv("EAGLE_BNECK_DS");
        const bool fused = B.prec == EAGLE_PREC_F32S && !(fe && atoi(fe) == 0) && bneck_supported(x, 64, 256);
        const bool dsf = fused && b == 0 && x.c == 64 && !(de && atoi(de) == 0);
        TView res = x;
        if (b == 0 && !dsf) res = B.conv(x, q + "downsample.0", q + "downsample.1", 1, 0, nullptr, nullptr, 0);
        if (fused) {
            TView y = dsf ? B.bottleneck(x, q, x, q + "downsample.0", q + "downsample.1") : B.bottleneck(x, q, res);
            if (b == 0 && !dsf) B.release(res);
            B.release(x);
            x = y;
            continue;
        }
        TView o1 = B.conv(x, q + "conv1", q + "bn1", 1, 0, nullptr, nullptr, R);
        TView o2 = B.conv(o1, q + "conv2", q + "bn2", 1, 0, nullptr, nullptr, R);
        TView y = B.conv(o2, q + "conv3", q + "bn3", 1, 0, &res, nullptr, R);
        B.release(o1); B.release(o2);
        if (b == 0) B.release(res);
        B.release(x);
        x = y;
    }
    std::vector<TView> ys(2);
    B.fork_join(Op::FORK, 1);                               // the two transition convolutions read the same tensor: side by side where branch streams are on
    B.set_stream(1);
    ys[1] = B.conv(x, P + "transition1.1.0.0", P + "transition1.1.0.1", 2, 0, nullptr, nullptr, R);
    B.set_stream(0);
    ys[0] = B.conv(x, P + "transition1.0.0", P + "transition1.0.1", 1, 0, nullptr, nullptr, R);
    B.fork_join(Op::JOIN, 1);
    B.release(x);
    ys = hr_stage(B, ys, 2, 1, 2, false);
    ys.push_back(B.conv(ys.back(), P + "transition2.2.0.0", P + "transition2.2.0.1", 2, 0, nullptr, nullptr, R));
    ys = hr_stage(B, ys, 3, 4, 3, false);
    ys.push_back(B.conv(ys.back(), P + "transition3.3.0.0", P + "transition3.3.0.1", 2, 0, nullptr, nullptr, R));
    ys = hr_stage(B, ys, 4, 3, 4, true);
    // fp16 family: sigmoid + per-tile maxima ride in the head convolution's epilogue (no logit tensor in HBM); the exact family
    // keeps the fp32 logits and heat_argmax_kernel
    ArgmaxPart* const* am = (prec_is_f16_kernels(B.prec) && !getenv("EAGLE_NO_FUSED_ARGMAX")) ? &B.H->cur_parts : nullptr;
    TView logits = B.conv(ys[0], "unnormalized_model.1", "", 1, 0, nullptr, nullptr, 0, nullptr, true, am);
    B.release(ys[0]);
    return logits;
}

// ------------------------------------------------------------------------------------------------------------
// YOLOv8 detect (ultralytics graph).  Concats are channel slices of one buffer: producers write
// straight into their slice, consumers read slices; nothing is copied except the nearest-x2 upsample.
// ------------------------------------------------------------------------------------------------------------
struct YoloDims { int c[5]; int n[4]; };
static YoloDims yolo_dims(int variant)
{
    static const double D[5] = {0.33, 0.33, 0.67, 1.0, 1.0}, Wd[5] = {0.25, 0.5, 0.75, 1.0, 1.25};
    static const int MC[5] = {1024, 1024, 768, 512, 512};
    YoloDims y;
    const int base[5] = {64, 128, 256, 512, 1024};
    for (int i = 0; i < 5; ++i) y.c[i] = (int)std::ceil(std::min(base[i], MC[variant]) * Wd[variant] / 8.0) * 8;
    const int rep[4] = {3, 6, 6, 3};
    for (int i = 0; i < 4; ++i) y.n[i] = std::max((int)std::nearbyint(rep[i] * D[variant]), 1);
    return y;
}

struct YoloBuilder {
    Builder& B;
    int S = ACT_SILU;
    TView cv(const TView& x, const std::string& name, int stride = 1, const TView* r1 = nullptr, const TView* out = nullptr)
    {
        return B.conv(x, name + ".conv", name + ".bn", stride, S, r1, nullptr, 0, out);
    }
    // C2f writing its result into `out` (or a fresh tensor)
    TView c2f(const TView& x, int idx, int cout, int n, bool shortcut, const TView* out = nullptr)
    {
        const std::string p = "model." + std::to_string(idx);
        const int c = cout / 2;
        TView cat = B.act(x.h, x.w, (2 + n) * c);
        TView first = cat.slice(0, 2 * c);
        cv(x, p + ".cv1", 1, nullptr, &first);
        for (int k = 0; k < n; ++k) {
            TView in = cat.slice((1 + k) * c, c);
            TView t = cv(in, p + ".m." + std::to_string(k) + ".cv1");
            TView o = cat.slice((2 + k) * c, c);
            cv(t, p + ".m." + std::to_string(k) + ".cv2", 1, shortcut ? &in : nullptr, &o);
            B.release(t);
        }
        TView y = cv(cat, p + ".cv2", 1, nullptr, out);
        B.release(cat);
        return y;
    }
};

// mixed (EAGLE_DET_PREC_MIXED; VERDICT r5 task 7): the trunk in the split family, the LAST C2f of every level (model.15 / 18 / 21), the two stride-2 convolutions between them
// (model.16 / 19) and Detect (model.22) in the exact fp32 family; the seam is an exact conversion of the three concat buffers (split_to_f32_launch).
static void build_yolo(Builder& B, const TView& x_in, int variant, DetLevel lv[3], int nc, bool mixed = false)
{
    YoloBuilder Y{B};
    const YoloDims d = yolo_dims(variant);
    const int c1 = d.c[0], c2 = d.c[1], c3 = d.c[2], c4 = d.c[3], c5 = d.c[4];
    TView x0 = Y.cv(x_in, "model.0", 2);
    TView x1 = Y.cv(x0, "model.1", 2); B.release(x0);
    TView x2 = Y.c2f(x1, 2, c2, d.n[0], true); B.release(x1);
    TView x3 = Y.cv(x2, "model.3", 2); B.release(x2);
    const int h3 = x3.h, w3 = x3.w;                                   // stride 8
    const int h4 = (h3 - 1) / 2 + 1, w4 = (w3 - 1) / 2 + 1, h5 = (h4 - 1) / 2 + 1, w5 = (w4 - 1) / 2 + 1;
    // concat buffers of the head, allocated up-front so the backbone can write P3/P4/P5 into their slices
    TView cat15 = B.act(h3, w3, c4 + c3);      // [up(h12) | p3]
    TView cat12 = B.act(h4, w4, c5 + c4);      // [up(p5)  | p4]
    TView cat18 = B.act(h4, w4, c3 + c4);      // [conv16  | h12]
    TView cat21 = B.act(h5, w5, c4 + c5);      // [conv19  | p5]
    TView p3 = cat15.slice(c4, c3), p4 = cat12.slice(c5, c4), p5 = cat21.slice(c4, c5), h12 = cat18.slice(c3, c4);
    Y.c2f(x3, 4, c3, d.n[1], true, &p3); B.release(x3);
    TView x5 = Y.cv(p3, "model.5", 2);
    Y.c2f(x5, 6, c4, d.n[2], true, &p4); B.release(x5);
    TView x7 = Y.cv(p4, "model.7", 2);
    TView x8 = Y.c2f(x7, 8, c5, d.n[3], true); B.release(x7);
    {   // SPPF
        const int ch = c5 / 2;
        TView cat = B.act(h5, w5, 4 * ch);
        TView s0 = cat.slice(0, ch), s1 = cat.slice(ch, ch), s2 = cat.slice(2 * ch, ch), s3 = cat.slice(3 * ch, ch);
        Y.cv(x8, "model.9.cv1", 1, nullptr, &s0); B.release(x8);
        B.other([s0, s1](hipStream_t s) { maxpool5_launch(s0, s1, s); }, "maxpool5", 2 * Builder::vbytes(s0));
        B.other([s1, s2](hipStream_t s) { maxpool5_launch(s1, s2, s); }, "maxpool5", 2 * Builder::vbytes(s0));
        B.other([s2, s3](hipStream_t s) { maxpool5_launch(s2, s3, s); }, "maxpool5", 2 * Builder::vbytes(s0));
        Y.cv(cat, "model.9.cv2", 1, nullptr, &p5);
        B.release(cat);
    }
    {
        TView u = cat12.slice(0, c5);
        B.other([p5, u](hipStream_t s) { upsample2_launch(p5, u, s); }, "upsample2", Builder::vbytes(p5) + Builder::vbytes(u));
        Y.c2f(cat12, 12, c4, d.n[0], false, &h12);
    }
    TView h15, h18, h21;
    {
        TView u = cat15.slice(0, c4);
        B.other([h12, u](hipStream_t s) { upsample2_launch(h12, u, s); }, "upsample2", Builder::vbytes(h12) + Builder::vbytes(u));
    }
    if (!mixed) {
        h15 = Y.c2f(cat15, 15, c3, d.n[0], false);
        {
            TView o = cat18.slice(0, c3);
            Y.cv(h15, "model.16", 2, nullptr, &o);
        }
        h18 = Y.c2f(cat18, 18, c4, d.n[0], false);
        {
            TView o = cat21.slice(0, c4);
            Y.cv(h18, "model.19", 2, nullptr, &o);
        }
        h21 = Y.c2f(cat21, 21, c5, d.n[0], false);
    } else {
        B.prec = EAGLE_PREC_F32;                           // every tensor allocated and every convolution built from here on: the exact family
        auto to_f32 = [&B](const TView& x, const TView& y) {
            B.other([x, y](hipStream_t s) { split_to_f32_launch(x, y, s); }, "split_to_f32", Builder::vbytes(x) + Builder::vbytes(y));
        };
        TView cat15f = B.act(h3, w3, c4 + c3), cat18f = B.act(h4, w4, c3 + c4), cat21f = B.act(h5, w5, c4 + c5);
        to_f32(cat15, cat15f);
        h15 = Y.c2f(cat15f, 15, c3, d.n[0], false);
        {
            TView o = cat18f.slice(0, c3);
            Y.cv(h15, "model.16", 2, nullptr, &o);
            to_f32(h12, cat18f.slice(c3, c4));
        }
        h18 = Y.c2f(cat18f, 18, c4, d.n[0], false);
        {
            TView o = cat21f.slice(0, c4);
            Y.cv(h18, "model.19", 2, nullptr, &o);
            to_f32(p5, cat21f.slice(c4, c5));
        }
        h21 = Y.c2f(cat21f, 21, c5, d.n[0], false);
    }
    const TView feats[3] = {h15, h18, h21};
    const float strides[3] = {8.f, 16.f, 32.f};
    int a0 = 0;
    for (int l = 0; l < 3; ++l) {
        const std::string pb = "model.22.cv2." + std::to_string(l), pc = "model.22.cv3." + std::to_string(l);
        TView b0 = Y.cv(feats[l], pb + ".0");
        TView b1 = Y.cv(b0, pb + ".1"); B.release(b0);
        TView box = B.conv(b1, pb + ".2", "", 1, 0, nullptr, nullptr, 0, nullptr, true); B.release(b1);
        TView k0 = Y.cv(feats[l], pc + ".0");
        TView k1 = Y.cv(k0, pc + ".1"); B.release(k0);
        TView cls = B.conv(k1, pc + ".2", "", 1, 0, nullptr, nullptr, 0, nullptr, true); B.release(k1);
        lv[l].box = box; lv[l].cls = cls; lv[l].gh = box.h; lv[l].gw = box.w; lv[l].stride = strides[l]; lv[l].a0 = a0;
        a0 += box.h * box.w;
    }
    (void)c1; (void)nc;
}

// ------------------------------------------------------------------------------------------------------------
// OSNet-x0.25 (appearance embeddings of the tracker; architecture table in eagle_amd/osnet.py, kernels in reid.hip).  The 1 x 1
// convolutions are ordinary Builder convolutions in the exact fp32 family; everything else is a reid_* launch.
// ------------------------------------------------------------------------------------------------------------
static const int REID_NB = 64;                  // crops per pass
static const char* RP = "reid.";

struct ReidFold { std::vector<float> scale, shift; };
static ReidFold reid_bn(Builder& B, const std::string& bn, int c)
{
    const HostTensor &g = B.W(bn + ".weight"), &b = B.W(bn + ".bias"), &m = B.W(bn + ".running_mean"), &v = B.W(bn + ".running_var");
    ReidFold f; f.scale.resize(c); f.shift.resize(c);
    for (int o = 0; o < c; ++o) {
        const double sc = (double)g.data[o] / std::sqrt((double)v.data[o] + 1e-5);
        f.scale[o] = (float)sc; f.shift[o] = (float)((double)b.data[o] - (double)m.data[o] * sc);
    }
    return f;
}

static TView reid_light(Builder& B, const TView& x, const std::string& lc, int mid)
{
    // LightConv3x3: 1x1 linear convolution -> depthwise 3x3 -> BatchNorm -> ReLU
    TView t = B.conv(x, lc + ".conv1", "", 1, 0, nullptr, nullptr, 0);
    const HostTensor& w = B.W(lc + ".conv2.weight");                       // [mid, 1, 3, 3]
    const ReidFold f = reid_bn(B, lc + ".bn", mid);
    const int C = t.c;
    std::vector<float> wk((size_t)9 * C, 0.f), bk(C, 0.f);
    for (int c = 0; c < mid; ++c) {
        for (int k = 0; k < 9; ++k) wk[(size_t)k * C + c] = (float)((double)w.data[(size_t)c * 9 + k] * (double)f.scale[c]);
        bk[c] = f.shift[c];
    }
    const float* dw = (const float*)B.net->upload(wk.data(), wk.size() * 4);
    const float* db = (const float*)B.net->upload(bk.data(), bk.size() * 4);
    TView y = B.act(t.h, t.w, C);
    const int n = B.N;
    B.other([t, dw, db, y, n](hipStream_t s) { reid_dw3_launch(t, dw, db, y, n, s); }, "reid dw3x3", 2 * Builder::vbytes(y));
    B.release(t);
    return y;
}

static TView reid_osblock(Builder& B, const TView& x, const std::string& b, int cin, int cout)
{
    const int mid = cout / 4, R = ACT_RELU;
    TView x1 = B.conv(x, b + ".conv1.conv", b + ".conv1.bn", 1, 0, nullptr, nullptr, R);
    TView st[4];
    st[0] = reid_light(B, x1, b + ".conv2a", mid);
    const char* names[3] = {"b", "c", "d"};
    for (int k = 0; k < 3; ++k) {
        TView y = x1;
        for (int d = 0; d < k + 2; ++d) {
            TView nx = reid_light(B, y, b + ".conv2" + names[k] + "." + std::to_string(d), mid);
            if (d > 0) B.release(y);
            y = nx;
        }
        st[k + 1] = y;
    }
    B.release(x1);
    // the shared ChannelGate and the four-stream sum
    const HostTensor &w1 = B.W(b + ".gate.fc1.weight"), &b1 = B.W(b + ".gate.fc1.bias"), &w2 = B.W(b + ".gate.fc2.weight"), &b2 = B.W(b + ".gate.fc2.bias");
    const int r = (int)w1.shape[0];
    const float* d1 = (const float*)B.net->upload(w1.data.data(), w1.data.size() * 4);
    const float* e1 = (const float*)B.net->upload(b1.data.data(), b1.data.size() * 4);
    const float* d2 = (const float*)B.net->upload(w2.data.data(), w2.data.size() * 4);
    const float* e2 = (const float*)B.net->upload(b2.data.data(), b2.data.size() * 4);
    const int C = st[0].c, n = B.N;
    float* g = (float*)B.net->get((size_t)n * 4 * C * 4);
    TView x2 = B.act(st[0].h, st[0].w, C);
    const TView s0 = st[0], s1 = st[1], s2 = st[2], s3 = st[3];
    B.other([s0, s1, s2, s3, d1, e1, d2, e2, mid, r, g, x2, n](hipStream_t s) { const TView ss[4] = {s0, s1, s2, s3}; reid_gate_launch(ss, d1, e1, d2, e2, mid, r, g, x2, n, s); },
            "reid gate", 5 * Builder::vbytes(x2));
    for (auto& t : st) B.release(t);
    TView ident = x;
    if (cin != cout) ident = B.conv(x, b + ".downsample.conv", b + ".downsample.bn", 1, 0, nullptr, nullptr, 0);
    TView y = B.conv(x2, b + ".conv3.conv", b + ".conv3.bn", 1, 0, &ident, nullptr, R);       // relu(conv3(x2) + identity): the residual add is commutative
    B.release(x2);
    if (cin != cout) B.release(ident);
    return y;
}

static void build_reid(Builder& B, EagleHandle* h)
{
    const std::string P = RP;
    const int n = B.N;
    h->reid_in = B.act(256, 128, 4);
    // stem: 7 x 7 / 2 convolution + BatchNorm + ReLU (weights [7][7][3][16] with the BN scale folded in), 3 x 3 / 2 max-pool
    const HostTensor& w1 = B.W(P + "conv1.conv.weight");                    // [16, 3, 7, 7]
    const ReidFold f1 = reid_bn(B, P + "conv1.bn", 16);
    std::vector<float> wk(7 * 7 * 3 * 16), bk(16);
    for (int o = 0; o < 16; ++o) {
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 49; ++k) wk[(size_t)(k * 3 + c) * 16 + o] = (float)((double)w1.data[((size_t)o * 3 + c) * 49 + k] * (double)f1.scale[o]);
        bk[o] = f1.shift[o];
    }
    const float* dw = (const float*)B.net->upload(wk.data(), wk.size() * 4);
    const float* db = (const float*)B.net->upload(bk.data(), bk.size() * 4);
    TView c1 = B.act(128, 64, 16);
    const TView in = h->reid_in;
    B.other([in, dw, db, c1, n](hipStream_t s) { reid_conv7_launch(in, dw, db, c1, n, s); }, "reid conv7x7", Builder::vbytes(in) + Builder::vbytes(c1));
    TView x = B.act(64, 32, 16);
    B.other([c1, x, n](hipStream_t s) { reid_maxpool3s2_launch(c1, x, n, s); }, "reid maxpool", Builder::vbytes(c1) + Builder::vbytes(x));
    B.release(c1);
    static const int CH[4] = {16, 64, 96, 128};
    const char* stage[3] = {"conv2", "conv3", "conv4"};
    for (int sI = 0; sI < 3; ++sI) {
        const int cin = CH[sI], cout = CH[sI + 1];
        const std::string sp = P + stage[sI];
        TView y = reid_osblock(B, x, sp + ".0", cin, cout); B.release(x); x = y;
        y = reid_osblock(B, x, sp + ".1", cout, cout); B.release(x); x = y;
        if (sI < 2) {                                       // transition: Conv1x1 + BN + ReLU, AvgPool2d(2, 2)
            TView t = B.conv(x, sp + ".2.0.conv", sp + ".2.0.bn", 1, 0, nullptr, nullptr, ACT_RELU); B.release(x);
            TView p = B.act(t.h / 2, t.w / 2, t.c);
            B.other([t, p, n](hipStream_t s) { reid_avgpool2_launch(t, p, n, s); }, "reid avgpool", Builder::vbytes(t) + Builder::vbytes(p));
            B.release(t);
            x = p;
        }
    }
    TView c5 = B.conv(x, P + "conv5.conv", P + "conv5.bn", 1, 0, nullptr, nullptr, ACT_RELU); B.release(x);
    // head: global average -> Linear(128, 512) + bias -> BatchNorm1d -> ReLU, the BN folded into the linear layer
    const HostTensor &fw = B.W(P + "fc.0.weight"), &fb = B.W(P + "fc.0.bias");
    const ReidFold ff = reid_bn(B, P + "fc.1", EAGLE_REID_DIM);
    std::vector<float> hw((size_t)EAGLE_REID_DIM * c5.c, 0.f), hb(EAGLE_REID_DIM);
    for (int o = 0; o < EAGLE_REID_DIM; ++o) {
        for (int c = 0; c < 128; ++c) hw[(size_t)o * c5.c + c] = (float)((double)fw.data[(size_t)o * 128 + c] * (double)ff.scale[o]);
        hb[o] = (float)((double)fb.data[o] * (double)ff.scale[o] + (double)ff.shift[o]);
    }
    const float* dhw = (const float*)B.net->upload(hw.data(), hw.size() * 4);
    const float* dhb = (const float*)B.net->upload(hb.data(), hb.size() * 4);
    h->reid_feats = (float*)B.net->get((size_t)n * EAGLE_REID_DIM * 4);
    float* feats = h->reid_feats;
    B.other([c5, dhw, dhb, feats, n](hipStream_t s) { reid_head_launch(c5, dhw, dhb, feats, EAGLE_REID_DIM, n, s); }, "reid head", Builder::vbytes(c5));
    h->reid_crops = (EagleCrop*)B.net->get(sizeof(EagleCrop) * (size_t)n);
    HIP_CHECK(hipHostMalloc((void**)&h->reid_crops_h, sizeof(EagleCrop) * (size_t)n, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void**)&h->reid_feats_h, sizeof(float) * EAGLE_REID_DIM * (size_t)n, hipHostMallocDefault));
}

// ------------------------------------------------------------------------------------------------------------
// step execution
// ------------------------------------------------------------------------------------------------------------
// profiling mode only: HIP events around one non-convolution launch on stream st (the stream the kernel is launched on)
static int ktab_index(EagleHandle* h, const char* name)
{
    for (size_t i = 0; i < h->ktab.size(); ++i) if (!strcmp(h->ktab[i].name, name)) return (int)i;
    EagleKernelTime e; memset(&e, 0, sizeof(e)); strncpy(e.name, name, sizeof(e.name) - 1);
    h->ktab.push_back(e);
    return (int)h->ktab.size() - 1;
}
template <class F>
static void timed(EagleHandle* h, const char* name, double bytes, hipStream_t st, F&& fn)
{
    if (!h->prof) { fn(); return; }
    while (h->span_pool.size() < h->span_used + 2) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); h->span_pool.push_back(e); }
    EagleHandle::Span sp{ktab_index(h, name), bytes, h->span_pool[h->span_used], h->span_pool[h->span_used + 1]};
    h->span_used += 2;
    HIP_CHECK(hipEventRecord(sp.a, st));
    fn();
    HIP_CHECK(hipEventRecord(sp.b, st));
    h->spans.push_back(sp);
}

static size_t sat_pad_bytes(int B) { return ((size_t)B * sizeof(unsigned) + 255) & ~(size_t)255; }

static void run_net(EagleHandle* h, Net* net, hipStream_t s, size_t& ev_i)
{
    const bool multi = h->multi_stream && !h->prof;
    for (Op& op : net->ops) {
        if (op.kind == Op::FORK) {
            if (multi) {
                HIP_CHECK(hipEventRecord(h->ev_fork, s));
                for (int k = 0; k < op.nbranch; ++k) HIP_CHECK(hipStreamWaitEvent(h->s_br[k], h->ev_fork, 0));
            }
            continue;
        }
        if (op.kind == Op::JOIN) {
            if (multi)
                for (int k = 0; k < op.nbranch; ++k) {
                    HIP_CHECK(hipEventRecord(h->ev_join[k], h->s_br[k]));
                    HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[k], 0));
                }
            continue;
        }
        hipStream_t st = (multi && op.stream > 0) ? h->s_br[op.stream - 1] : s;
        if ((g_dbg_skip & 64) && op.kind == Op::OTHER) continue;
        if ((g_dbg_skip & 128) && op.kind == Op::CONV) continue;
        if (h->prof && op.kind == Op::CONV) {
            HIP_CHECK(hipEventRecord(h->conv_ev[ev_i++], st));
            op.run(st);
            HIP_CHECK(hipEventRecord(h->conv_ev[ev_i++], st));
            h->conv_ops.push_back(&op);
        } else {
            timed(h, op.tag, op.bytes, st, [&] { op.run(st); });
        }
    }
}

// networks + decode/NMS + heat-map maxima of one batch (buffers of parity p), reading frames from d_src (device)
static void enqueue_compute(EagleHandle* h, int p, const uint8_t* d_src, int n_active)
{
    const EagleConfig& c = h->cfg;
    const int B = c.batch;
    EagleHandle::StepBuf& sb = h->sb[p];
    size_t ev_i = 0;
    HIP_CHECK(hipMemsetAsync(sb.d_sat, 0, sat_pad_bytes(B) + sizeof(EagleFrameResult) * B, h->s_main));      // saturation words + records
    h->cur_sat = sb.d_sat;
    const double esz = h->prec == EAGLE_PREC_F16 ? 2 : 4;
    if (!(g_dbg_skip & 16))
        timed(h, "preprocess", (double)n_active * ((double)c.frame_h * c.frame_w * 3 + (540.0 * 960 + (double)h->lb.out_h * h->lb.out_w) * h->kp_in.c * esz), h->s_main,
              [&] {
                  preprocess_launch(h->prec, d_src, n_active, c.frame_h, c.frame_w, h->kp_in, h->det_in, h->lb, h->s_main, 3, h->det_prec);      // one launch, each tensor in its network's format
              });
    const bool two = !h->prof;
    hipStream_t sd = two ? h->s_det : h->s_main;
    if (two) {
        HIP_CHECK(hipEventRecord(h->ev_pre, h->s_main));
        HIP_CHECK(hipStreamWaitEvent(sd, h->ev_pre, 0));
    }
    if (!(g_dbg_skip & 2)) run_net(h, h->yo.get(), sd, ev_i);            // detector branch
    if (!(g_dbg_skip & 4)) {
        // (64 box logits + 16-padded class logits) fp32 in, 4 box floats + confidence + class + sort key out, per anchor
        timed(h, "yolo_decode", (double)B * h->ds.A * ((64 + 16) * 4.0 + 4 * 4 + 4 + 4 + 8), sd, [&] { yolo_decode_launch(h->levels, 3, B, 5, c.detector_floor, h->ds, sd); });
        timed(h, "nms", (double)B * h->ds.A * 8.0, sd, [&] { nms_launch(h->ds, B, h->pp, sb.d_out, sd); });      // the key array, read once
    }
    if (two) HIP_CHECK(hipEventRecord(h->ev_det, sd));
    h->cur_parts = sb.parts;
    if (!(g_dbg_skip & 1)) run_net(h, h->hr.get(), h->s_main, ev_i);     // keypoint branch
    if (!(g_dbg_skip & 32) && !h->fused_argmax)
        timed(h, "heat_argmax", (double)h->logits.n * h->logits.h * h->logits.w * h->logits.cs * 4.0, h->s_main, [&] { heat_argmax_launch(h->logits, sb.parts, h->hm_chunks, h->s_main); });
    if (two) HIP_CHECK(hipStreamWaitEvent(h->s_main, h->ev_det, 0));     // join
}

// use_graph: 1 = every step is replayed; 2 = only inside calls of at least three steps — there the graph launch of step i + 1 hides behind step i on the GPU and saves the host
// the 383 launches, while a ONE-step call of a large batch pays the graph launch in full before anything runs (B = 25 per call: 658 -> 455 frames/s,
// profiles/r05c_latency_modes.txt).  Measured at batch 50, 20 steps: nothing on /opt/rocm's runtime (765.3 against 766.1 frames/s), +1.6 % on the PyTorch wheel's ROCm 7.0.2 runtime,
// whose launch path is slower (739 -> 751): bench.py asks for 2 in its multi-rank path; "auto" stays at plain launches for batch > EAGLE_SMALL_BATCH (a capture costs ~80 ms per
// (slot, frame count), which a short first call would pay inside its own latency)
static bool graph_on(const EagleHandle* h) { return !h->graph_broken && (h->cfg.use_graph == 1 || (h->cfg.use_graph == 2 && h->call_steps >= 3)); }

static void launch_step(EagleHandle* h, int p, const uint8_t* d_src, int n_active)
{
    const EagleConfig& c = h->cfg;
    EagleHandle::StepBuf& sb = h->sb[p];
    bool replayed = false;
    if (graph_on(h) && !h->prof) {
        if (!h->warmed) {   // first call eager: lets every launcher set its function attributes outside a capture
            enqueue_compute(h, p, d_src, n_active);
            HIP_CHECK(hipStreamSynchronize(h->s_main));
            h->warmed = true;
        }
        if (d_src != sb.d_frames) fail(EAGLE_E_STATE, "graph replay needs the slot's staging buffer as the source");
        auto it = sb.graphs.find(n_active);
        if (it == sb.graphs.end() && sb.capture_skip > 0) --sb.capture_skip;
        else if (it == sb.graphs.end()) {
            CaptureExclusive only_this_thread_talks_to_hip;
            if (only_this_thread_talks_to_hip.ok) {
                // thread-local capture mode: only THIS thread is held to capture-safe calls while the capture is open (it makes none: the warm-up step above has
                // set every function attribute and allocated the zero / trash pages).  The global mode made every hipMalloc / hipFree / synchronise of ANY other
                // thread fail with "operation not permitted when stream is capturing" — a second handle on another host thread, which is how the concurrency
                // tests and a multi-handle server run — and left this stream in a broken capture (round 5: six tests of the suite, once small batches captured by default)
                struct CaptureScope {                       // a throw between Begin and End must not leave s_main capturing: end it, drop the graph, stop replaying
                    EagleHandle* h; bool open = false; hipGraph_t g = nullptr;
                    ~CaptureScope()
                    {
                        if (open) { (void)hipStreamEndCapture(h->s_main, &g); (void)hipGetLastError(); h->graph_broken = true; }
                        if (g) (void)hipGraphDestroy(g);
                    }
                } cs{h};
                HIP_CHECK(hipStreamBeginCapture(h->s_main, hipStreamCaptureModeThreadLocal));
                cs.open = true;
                enqueue_compute(h, p, d_src, n_active);
                cs.open = false;
                HIP_CHECK(hipStreamEndCapture(h->s_main, &cs.g));
                hipGraphExec_t ge = nullptr;
                if (hipGraphInstantiate(&ge, cs.g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); h->graph_broken = true; fail(EAGLE_E_HIP, "hipGraphInstantiate failed"); }
                if (sb.graphs.size() >= 16) { (void)hipGraphExecDestroy(sb.graphs.begin()->second); sb.graphs.erase(sb.graphs.begin()); }      // (frame counts 1 .. batch: bounded)
                it = sb.graphs.emplace(n_active, ge).first;
                sb.capture_backoff = 0;
                ++h->graph_captures;
            } else {
                sb.capture_backoff = std::min(64, std::max(1, sb.capture_backoff * 2));
                sb.capture_skip = sb.capture_backoff;
                ++h->graph_skipped;
            }
        }
        if (it != sb.graphs.end()) { HIP_CHECK(hipGraphLaunch(it->second, h->s_main)); replayed = true; }
    }
    if (!replayed) enqueue_compute(h, p, d_src, n_active);
    hipStream_t sp = h->prof ? h->s_main : h->s_post;
    if (!h->prof) {
        HIP_CHECK(hipEventRecord(sb.ev_compute, h->s_main));
        HIP_CHECK(hipStreamWaitEvent(sp, sb.ev_compute, 0));
    }
    if (!(g_dbg_skip & 8)) timed(h, "post (geometry)", 0, sp, [&] { post_launch(sb.parts, c.batch, h->pp, sb.d_out, sp); });
    HIP_CHECK(hipMemcpyAsync(sb.h_sat, sb.d_sat, sat_pad_bytes(c.batch) + sizeof(EagleFrameResult) * n_active, hipMemcpyDeviceToHost, sp));
    HIP_CHECK(hipEventRecord(sb.ev_done, sp));
}

static void collect_step(EagleHandle* h, int p, int n_active, EagleFrameResult* out)
{
    HIP_CHECK(hipEventSynchronize(h->sb[p].ev_done));
    memcpy(out, h->sb[p].h_out, sizeof(EagleFrameResult) * n_active);
    for (int i = 0; i < n_active; ++i)                      // f32s: frames in which an activation left the split format's range (|v| > 4094) are flagged
        if (h->sb[p].h_sat[i]) { out[i].pad[1] = 1; h->sat_events += h->sb[p].h_sat[i]; ++h->sat_frames; }
    h->timings.n_launches += h->n_launch;
    h->timings.n_conv_launches += h->n_conv;
    h->timings.conv_flop += h->conv_flop_step;
    if (h->prof) {
        for (size_t i = 0; i + 1 < h->conv_ev.size(); i += 2) {
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, h->conv_ev[i], h->conv_ev[i + 1]));
            h->timings.conv_ms += t;
            if (i / 2 < h->conv_ops.size()) {
                const Op* op = h->conv_ops[i / 2];
                EagleKernelTime& e = h->ktab[ktab_index(h, op->tag)];
                e.ms += t; e.launches += 1; e.bytes += op->bytes; e.flop += op->flop;
            }
        }
        h->conv_ops.clear();
        for (const EagleHandle::Span& sp : h->spans) {
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, sp.a, sp.b));
            EagleKernelTime& e = h->ktab[sp.k];
            e.ms += t; e.launches += 1; e.bytes += sp.bytes;
        }
        h->spans.clear(); h->span_used = 0;
    }
}

// all batches of one call; src_of(i) yields the device pointer of batch i's frames (after any staging copy)
template <class Stage>
static void run_pipeline(EagleHandle* h, int n, EagleFrameResult* out, Stage stage)
{
    const int B = h->cfg.batch;
    memset(&h->timings, 0, sizeof(h->timings));
    h->sat_events = 0; h->sat_frames = 0;
    if (n == 0) return;
    h->call_steps = (n + B - 1) / B;
    HIP_CHECK(hipEventRecord(h->ev_t0, h->s_main));
    int prev_n = 0, prev_i = 0, k = 0;
    for (int i = 0; i < n; i += B, ++k) {
        const int p = k & 1, na = std::min(B, n - i);
        const uint8_t* src = stage(p, i, na);
        launch_step(h, p, src, na);
        if (k > 0) collect_step(h, p ^ 1, prev_n, out + prev_i);
        prev_n = na; prev_i = i;
        if (h->prof) { collect_step(h, p, na, out + i); prev_n = 0; }     // profiling mode: strictly serial
    }
    if (prev_n > 0) collect_step(h, (k - 1) & 1, prev_n, out + prev_i);
    hipStream_t sp = h->prof ? h->s_main : h->s_post;
    HIP_CHECK(hipEventRecord(h->ev_t1, sp));
    HIP_CHECK(hipEventSynchronize(h->ev_t1));
    HIP_CHECK(hipEventSynchronize(h->ev_t0));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
    h->timings.total_ms = ms;
    h->timings.sat_events = (int32_t)std::min<long long>(h->sat_events, 0x7fffffff);
    h->timings.sat_frames = h->sat_frames;
}

// EAGLE_PREC_F32S stores clip at +-4094 instead of overflowing; a call in which that happened must not look like a success
static void check_saturation(EagleHandle* h, const char* what)
{
    if (h->sat_events > 0 && !h->cfg.allow_saturation)
        fail(EAGLE_E_RANGE, "%s: %lld activation stores in %d frame(s) left the range of the f32s tensor format (|v| > 4094) and were clipped; the records are "
             "written (EagleFrameResult.pad[1] marks the frames) but are not fp32-grade.  Use EAGLE_PREC_F32 for these weights, or set EagleConfig.allow_saturation",
             what, h->sat_events, h->sat_frames);
}

// ---- clip session (optical-flow cadence) ----------------------------------------------------------------------------------
// Three streams: s_det runs the detector pass, s_main the HRNet pass (+ gray pyramids, operator calls), s_post the sequential
// loop body (K12 + K13 per frame).  The passes of later frames overlap the loop of earlier ones; events order them.
static void clip_sync(EagleHandle* h)
{
    HIP_CHECK(hipStreamSynchronize(h->s_det));
    HIP_CHECK(hipStreamSynchronize(h->s_main));
    HIP_CHECK(hipStreamSynchronize(h->s_post));
}

static void clip_close(EagleHandle* h)
{
    EagleHandle::Clip& c = h->clip;
    if (c.open) { (void)hipStreamSynchronize(h->s_det); (void)hipStreamSynchronize(h->s_main); (void)hipStreamSynchronize(h->s_post); }
    for (auto& p : c.g) { if (p) (void)hipFree(p); p = nullptr; }
    if (c.recs) (void)hipFree(c.recs);
    if (c.mem) (void)hipFree(c.mem);
    if (c.st) (void)hipFree(c.st);
    if (c.st_op) (void)hipFree(c.st_op);
    if (c.h_st) (void)hipHostFree(c.h_st);
    if (c.h_zero) (void)hipHostFree(c.h_zero);
    if (c.h_tail) (void)hipHostFree(c.h_tail);
    if (c.h_mem) (void)hipHostFree(c.h_mem);
    if (c.ecc_small) (void)hipFree(c.ecc_small);
    if (c.ecc_pairs) (void)hipFree(c.ecc_pairs);
    if (c.ecc_out) (void)hipFree(c.ecc_out);
    for (hipEvent_t e : {c.ev_gray, c.ev_det, c.ev_kp, c.ev_loop}) if (e) (void)hipEventDestroy(e);
    c = EagleHandle::Clip();
}

static void clip_open(EagleHandle* h, const uint8_t* d_bgr, int n)
{
    clip_close(h);
    const EagleConfig& cf = h->cfg;
    EagleHandle::Clip& c = h->clip;
    c.cv.bgr = d_bgr; c.cv.n = n; c.cv.h = cf.frame_h; c.cv.w = cf.frame_w;
    c.cv.lh[0] = cf.frame_h; c.cv.lw[0] = cf.frame_w; c.cv.levels = 0;
    for (int l = 1; l <= 2; ++l) {                       // cv2 maxLevel = 2 (cm.py:65); a level must exceed the 15x15 window
        c.cv.lh[l] = (c.cv.lh[l - 1] + 1) / 2; c.cv.lw[l] = (c.cv.lw[l - 1] + 1) / 2;
        if (c.cv.lw[l] <= 15 || c.cv.lh[l] <= 15) break;
        c.cv.levels = l;
    }
    c.open = true;
    for (int l = 0; l < 3; ++l) {
        HIP_CHECK(hipMalloc((void**)&c.g[l], std::max<size_t>((size_t)n * c.cv.lh[l] * c.cv.lw[l], 16)));
        c.cv.g[l] = c.g[l];
    }
    const size_t nn = (size_t)std::max(n, 1);
    HIP_CHECK(hipMalloc((void**)&c.recs, sizeof(EagleFrameResult) * nn));
    HIP_CHECK(hipMalloc((void**)&c.mem, sizeof(MemList) * nn));
    HIP_CHECK(hipMalloc((void**)&c.st, sizeof(ChainState)));
    HIP_CHECK(hipMalloc((void**)&c.st_op, sizeof(ChainState)));
    HIP_CHECK(hipHostMalloc((void**)&c.h_st, sizeof(ChainState), hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void**)&c.h_zero, sizeof(ChainState), hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void**)&c.h_tail, sizeof(int) * 4, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void**)&c.h_mem, sizeof(MemList), hipHostMallocDefault));
    memset(c.h_zero, 0, sizeof(ChainState));
    c.h_zero->stalled = -1;
    c.h_tail[0] = -1; c.h_tail[1] = 0; c.h_tail[2] = -1;
    for (hipEvent_t* e : {&c.ev_gray, &c.ev_det, &c.ev_kp, &c.ev_loop}) HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    HIP_CHECK(hipMemsetAsync(c.mem, 0xFF, sizeof(MemList) * nn, h->s_main));                  // n = -1 everywhere
    HIP_CHECK(hipMemsetAsync(c.recs, 0, sizeof(EagleFrameResult) * nn, h->s_main));
    HIP_CHECK(hipMemsetAsync(h->clip_sat, 0, sat_pad_bytes(cf.batch), h->s_main));
    HIP_CHECK(hipMemcpyAsync(c.st, c.h_zero, sizeof(ChainState), hipMemcpyHostToDevice, h->s_main));
    if (n > 0) gray_pyramid_launch(d_bgr, n, c.cv.h, c.cv.w, c.g[0], c.g[1], c.g[2], h->s_main);
    HIP_CHECK(hipEventRecord(c.ev_gray, h->s_main));
    HIP_CHECK(hipStreamWaitEvent(h->s_det, c.ev_gray, 0));       // the record memset precedes the first detector write
    HIP_CHECK(hipEventRecord(c.ev_det, h->s_det));
    HIP_CHECK(hipEventRecord(c.ev_kp, h->s_main));
    HIP_CHECK(hipEventRecord(c.ev_loop, h->s_post));
}

// detector + decode + NMS + object rules of frames [first, first+count) (cm.py:331 detect_objects), records kept in HBM; asynchronous
static void clip_detect_objects(EagleHandle* h, int first, int count)
{
    const EagleConfig& cf = h->cfg;
    EagleHandle::Clip& c = h->clip;
    const int B = cf.batch;
    const size_t fb = (size_t)cf.frame_h * cf.frame_w * 3;
    size_t ev_i = 0;
    EagleHandle::StepBuf& sb = h->sb[0];
    const bool prof = h->prof; h->prof = false;
    h->cur_sat = h->clip_sat;
    // The passes of later chunks run under the sequential loop of earlier frames (three streams).  Round 1 had to serialise them
    // behind the loop because K12 was not reproducible next to the convolution kernels; the cause was the packed-fp32 code hipcc's
    // SLP vectoriser generated for K12 (Makefile), not the overlap.
    for (int i = first; i < first + count; i += B) {
        const int na = std::min(B, first + count - i);
        HIP_CHECK(hipMemsetAsync(sb.d_out, 0, sizeof(EagleFrameResult) * B, h->s_det));
        preprocess_launch(h->det_prec, c.cv.bgr + (size_t)i * fb, na, cf.frame_h, cf.frame_w, h->kp_in, h->det_in, h->lb, h->s_det, 2);
        run_net(h, h->yo.get(), h->s_det, ev_i);
        yolo_decode_launch(h->levels, 3, B, 5, cf.detector_floor, h->ds, h->s_det);
        nms_launch(h->ds, B, h->pp, sb.d_out, h->s_det);
        HIP_CHECK(hipMemcpyAsync(c.recs + i, sb.d_out, sizeof(EagleFrameResult) * na, hipMemcpyDeviceToDevice, h->s_det));
    }
    h->prof = prof;
    HIP_CHECK(hipEventRecord(c.ev_det, h->s_det));
}

// HRNet + heat-map maxima + decode of frames first, first+stride, ... -> mem[]; asynchronous
static void clip_detect_keypoints(EagleHandle* h, int first, int stride, int count)
{
    const EagleConfig& cf = h->cfg;
    EagleHandle::Clip& c = h->clip;
    const int B = cf.batch;
    const size_t fb = (size_t)cf.frame_h * cf.frame_w * 3;
    size_t ev_i = 0;
    EagleHandle::StepBuf& sb = h->sb[0];
    const bool prof = h->prof; h->prof = false;
    h->cur_sat = h->clip_sat;
    for (int k0 = 0; k0 < count; k0 += B) {
        const int na = std::min(B, count - k0);
        const uint8_t* src;
        if (stride == 1) src = c.cv.bgr + (size_t)(first + k0) * fb;
        else {
            for (int k = 0; k < na; ++k)
                HIP_CHECK(hipMemcpyAsync(sb.d_frames + (size_t)k * fb, c.cv.bgr + (size_t)(first + (k0 + k) * stride) * fb, fb, hipMemcpyDeviceToDevice, h->s_main));
            src = sb.d_frames;
        }
        preprocess_launch(h->prec, src, na, cf.frame_h, cf.frame_w, h->kp_in, h->det_in, h->lb, h->s_main, 1);
        h->cur_parts = sb.parts;
        run_net(h, h->hr.get(), h->s_main, ev_i);
        if (!h->fused_argmax) heat_argmax_launch(h->logits, sb.parts, h->hm_chunks, h->s_main);
        decode_mem_launch(sb.parts, na, h->pp, c.mem, first + k0 * stride, stride, h->s_main);
    }
    h->prof = prof;
    HIP_CHECK(hipEventRecord(c.ev_kp, h->s_main));
}

static void finalize(EagleHandle* h)
{
    const EagleConfig& c = h->cfg;
    const int B = c.batch;
    h->prec = c.precision;
    const bool det_mixed = c.det_precision == EAGLE_DET_PREC_MIXED;
    h->det_prec = det_mixed ? EAGLE_PREC_F32S : c.det_precision ? c.det_precision - 1 : c.precision;
    h->hr.reset(new Net); h->yo.reset(new Net); h->misc.reset(new Net);
    const int cin_pad = h->prec == EAGLE_PREC_F32 ? 4 : 8, det_cin_pad = h->det_prec == EAGLE_PREC_F32 ? 4 : 8;
    h->lb = letterbox_geometry(c.frame_h, c.frame_w, c.det_imgsz, c.letterbox);
    // inputs (written by the preprocess kernel)
    Builder Bh{h, h->hr.get(), h->prec, 1e-5, B};
    Builder By{h, h->yo.get(), h->det_prec, 1e-3, B};
    By.label_suffix = " d";
    h->kp_in = Bh.act(540, 960, cin_pad);
    h->det_in = By.act(h->lb.out_h, h->lb.out_w, det_cin_pad);
    h->logits = build_hrnet(Bh, h->kp_in);
    build_yolo(By, h->det_in, c.det_variant, h->levels, 5, det_mixed);
    if (h->weights.count(std::string(RP) + "conv1.conv.weight")) {          // appearance embeddings for the tracker: only when the caller loaded an OSNet
        h->reid.reset(new Net);
        Builder Br{h, h->reid.get(), EAGLE_PREC_F32, 1e-5, REID_NB};
        build_reid(Br, h);
    }
    // scratch
    Net* m = h->misc.get();
    for (auto& sb : h->sb) {
        sb.d_frames = (uint8_t*)m->get((size_t)B * c.frame_h * c.frame_w * 3);
        sb.parts = (ArgmaxPart*)m->get(sizeof(ArgmaxPart) * (size_t)B * h->hm_chunks * 64);
        const size_t sat_pad = sat_pad_bytes(B);
        sb.d_sat = (unsigned*)m->get(sat_pad + sizeof(EagleFrameResult) * (size_t)B);
        sb.d_out = (EagleFrameResult*)((char*)sb.d_sat + sat_pad);
        HIP_CHECK(hipHostMalloc((void**)&sb.h_sat, sat_pad + sizeof(EagleFrameResult) * (size_t)B, hipHostMallocDefault));
        sb.h_out = (EagleFrameResult*)((char*)sb.h_sat + sat_pad);
        HIP_CHECK(hipEventCreateWithFlags(&sb.ev_compute, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sb.ev_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sb.ev_copy, hipEventDisableTiming));
    }
    h->clip_sat = (unsigned*)m->get(sat_pad_bytes(B));
    HIP_CHECK(hipHostMalloc((void**)&h->clip_sat_h, sat_pad_bytes(B), hipHostMallocDefault));
    int A = 0;
    for (int l = 0; l < 3; ++l) A += h->levels[l].gh * h->levels[l].gw;
    h->ds.A = A;
    h->ds.boxes = (float*)m->get(sizeof(float) * 4 * (size_t)B * A);
    h->ds.conf = (float*)m->get(sizeof(float) * (size_t)B * A);
    h->ds.cls = (int*)m->get(sizeof(int) * (size_t)B * A);
    h->ds.keys = (unsigned long long*)m->get(sizeof(unsigned long long) * (size_t)B * A);
    h->ds.count = (int*)m->get(sizeof(int) * (size_t)B);
    PostParams& pp = h->pp;
    pp.frame_h = c.frame_h; pp.frame_w = c.frame_w; pp.in_h = h->lb.out_h; pp.in_w = h->lb.out_w;
    pp.hm_h = h->logits.h; pp.hm_w = h->logits.w; pp.hm_chunks = h->hm_chunks;
    pp.keypoint_conf = c.keypoint_conf; pp.detector_conf = c.detector_conf; pp.nms_iou = c.nms_iou;
    pp.ransac_thresh = c.ransac_thresh; pp.ransac_max_iters = c.ransac_max_iters; pp.lm_iters = c.lm_iters;
    h->n_conv = 0; h->conv_flop_step = 0; h->n_launch = 6;
    for (Net* n : {h->hr.get(), h->yo.get()})
        for (Op& op : n->ops) { ++h->n_launch; if (op.kind == Op::CONV) { ++h->n_conv; h->conv_flop_step += op.flop; } }
    h->conv_ev.resize((size_t)h->n_conv * 2);
    for (auto& e : h->conv_ev) HIP_CHECK(hipEventCreate(&e));
    h->weights.clear();      // host copies are no longer needed
    h->finalized = true;
}

}  // namespace eagle

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
#define API_BEGIN try { eagle::ApiGuard api_guard_;
#define API_END(h)                                                       \
    }                                                                    \
    catch (const eagle::Err& e) { if (h) (h)->err = e.msg; else eagle::g_create_error = e.msg; return e.code; } \
    catch (const std::exception& e) { if (h) (h)->err = e.what(); else eagle::g_create_error = e.what(); return EAGLE_E_INVALID; } \
    return EAGLE_OK;

extern "C" {

int eagle_abi_sizes(int32_t* o)
{
    if (!o) return EAGLE_E_INVALID;
    o[0] = (int32_t)sizeof(EagleConfig); o[1] = (int32_t)sizeof(EagleFrameResult); o[2] = (int32_t)sizeof(EagleDet); o[3] = (int32_t)sizeof(EagleKeypoint);
    return EAGLE_OK;
}

int eagle_default_config(EagleConfig* cfg)
{
    if (!cfg) return EAGLE_E_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0; cfg->frame_h = 720; cfg->frame_w = 1280;
    cfg->det_variant = EAGLE_DET_N; cfg->det_imgsz = 640; cfg->batch = 8; cfg->precision = EAGLE_PREC_F32S;      // fp32-grade results by default (the reference computes in fp32)
    cfg->keypoint_conf = 0.3; cfg->detector_conf = 0.35; cfg->ransac_thresh = 5.0;
    cfg->detector_floor = 0.15f; cfg->nms_iou = 0.7f;
    cfg->ransac_max_iters = 2000; cfg->lm_iters = 10;
    cfg->use_graph = EAGLE_AUTO; cfg->multi_stream = EAGLE_AUTO;      // small-batch mode (batch <= EAGLE_SMALL_BATCH): hipGraph replay + HRNet's branches on their own streams
    // "auto" (resolved by eagle_create from the `precision` the caller ends up with): next to split-family key-points the detector (1.4 % of the
    // FLOP with yolov8n) runs in the exact fp32 family — boxes, confidences, classes, the NMS order and therefore every detection-index id
    // (cm.py:598-627) equal the fp32 oracle's bit for bit; next to any other `precision` it runs in that same family
    cfg->det_precision = EAGLE_DET_PREC_AUTO;
    return EAGLE_OK;
}

int eagle_resolve_config(EagleConfig* cfg)
{
    if (!cfg) return EAGLE_E_INVALID;
    if (cfg->det_precision == EAGLE_DET_PREC_AUTO) cfg->det_precision = cfg->precision == EAGLE_PREC_F32S ? EAGLE_PREC_F32 + 1 : 0;
    // Small-batch mode (round 5; the reference's caller hands over ONE frame per iteration, cm.py:277).  A step of <= EAGLE_SMALL_BATCH frames leaves most
    // of the chip idle inside every launch (48->48 @135x240 is 116 workgroups per frame for 512 slots) and its 383 launches cost as much as its kernels:
    // the network phase is replayed as ONE hipGraph and HRNet's branches (and the detector) run on their own streams so that the small launches of
    // different branches fill the CUs together.  Measured on MI355X (default handle, per call incl. H2D and records back, bench.py `latency`): B = 1
    // 13.9 -> 9.6 ms, B = 4 252 -> 344 frames/s, B = 8 395 -> 505 frames/s; at B = 50 the same switches measure nothing (DESIGN.md §4b xii), so larger
    // batches keep plain launches (no graph replay).
    // Sweep on one box (profiles/r05c_latency_modes.txt; frames/s plain -> small-batch mode): B = 1 72 -> 105, 4 252 -> 344, 8 395 -> 505, 12 506 -> 571,
    // 16 530 -> 599 (the branch streams alone; the graph adds nothing beyond B = 8), 25 658 -> 674 (graph replay of a 25-frame step: 455, it loses), 50 0.
    if (cfg->use_graph == EAGLE_AUTO) cfg->use_graph = (cfg->batch >= 1 && cfg->batch <= EAGLE_SMALL_BATCH) ? 1 : 0;      // (2 = replay only inside calls of >= 3 steps: on request)
    // Branch streams: on for EVERY batch since round 6.  Until then they paid up to 16 frames per step and measured nothing at 50; with the fuse outputs on parallel
    // streams and the branches' blocks interleaved (hr_stage) the small launches of a module's fuse phase and the tails of its branch launches overlap at any
    // batch: B = 50, same box, three alternating pairs 810.2 / 806.4 / 805.7 -> 823.1 / 820.2 / 818.0 frames/s (profiles/r06aj_*).  EAGLE_MULTI_STREAM=0 in the
    // environment resolves "auto" to one stream per network (developer A/B).
    if (cfg->multi_stream == EAGLE_AUTO) cfg->multi_stream = (getenv("EAGLE_MULTI_STREAM") && atoi(getenv("EAGLE_MULTI_STREAM")) == 0) ? 0 : 1;
    return EAGLE_OK;
}

int eagle_create(const EagleConfig* cfg, EagleHandle** out)
{
    EagleHandle* h = nullptr;
    API_BEGIN
    if (!cfg || !out) fail(EAGLE_E_INVALID, "null argument");
    if (cfg->batch < 1 || cfg->frame_h < 32 || cfg->frame_w < 32) fail(EAGLE_E_INVALID, "bad batch/frame size");
    if (cfg->precision != EAGLE_PREC_F16 && cfg->precision != EAGLE_PREC_F32 && cfg->precision != EAGLE_PREC_F32S) fail(EAGLE_E_INVALID, "bad precision");
    if (cfg->det_variant < 0 || cfg->det_variant > 4) fail(EAGLE_E_INVALID, "bad detector variant");
    if (cfg->letterbox != EAGLE_LETTERBOX_RECT && cfg->letterbox != EAGLE_LETTERBOX_SQUARE) fail(EAGLE_E_INVALID, "letterbox: 0 (rect, auto=True) or 1 (square, auto=False)");
    if (cfg->det_imgsz < 32 || cfg->det_imgsz % 32) fail(EAGLE_E_INVALID, "det_imgsz must be a positive multiple of 32 (the detector's largest stride)");
    if (cfg->det_precision < EAGLE_DET_PREC_AUTO || cfg->det_precision > EAGLE_DET_PREC_MIXED) fail(EAGLE_E_INVALID, "bad detector precision");
    if (cfg->use_graph < EAGLE_AUTO || cfg->use_graph > 2 || cfg->multi_stream < EAGLE_AUTO || cfg->multi_stream > 1) fail(EAGLE_E_INVALID, "use_graph: -1 (auto), 0, 1 or 2; multi_stream: -1 (auto), 0 or 1");
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) fail(EAGLE_E_HIP, "device %d not present (%d visible)", cfg->device, ndev);
    HIP_CHECK(hipSetDevice(cfg->device));
    EagleHandle* nh = new EagleHandle;
    nh->cfg = *cfg;
    eagle_resolve_config(&nh->cfg);                        // det_precision "auto" -> a family, from the precision the caller actually asked for
    HIP_CHECK(hipStreamCreateWithFlags(&nh->s_main, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&nh->s_det, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&nh->s_post, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&nh->s_copy, hipStreamNonBlocking));
    for (auto& st : nh->s_br) HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&nh->ev_fork, hipEventDisableTiming));
    for (auto& e : nh->ev_join) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    nh->multi_stream = nh->cfg.multi_stream != 0;         // concurrent HRNet branches: pays for small batches only (eagle_resolve_config)
    HIP_CHECK(hipEventCreateWithFlags(&nh->ev_pre, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&nh->ev_det, hipEventDisableTiming));
    HIP_CHECK(hipEventCreate(&nh->ev_t0));
    HIP_CHECK(hipEventCreate(&nh->ev_t1));
    *out = nh;
    API_END(h)
}

void eagle_destroy(EagleHandle* h)
{
    eagle::ApiGuard not_during_a_capture;                  // hipDeviceSynchronize / hipFree below would invalidate a capture that another handle's thread has open
    if (h && h->clip.open) { (void)hipSetDevice(h->cfg.device); eagle::clip_close(h); }
    if (!h) return;
    if (h->tracker) eagle::tracker_destroy(h->tracker);
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    if (h->comm && h->rccl) {                               // the communicator of eagle_comm_init: released with the handle, before its streams go
        typedef int (*fn_comm_destroy)(void*);
        if (fn_comm_destroy cd = (fn_comm_destroy)dlsym(h->rccl, "ncclCommDestroy")) (void)cd(h->comm);
        h->comm = nullptr;
    }
    for (auto& sb : h->sb) {
        for (auto& kv : sb.graphs) (void)hipGraphExecDestroy(kv.second);
        if (sb.h_sat) (void)hipHostFree(sb.h_sat);       // (h_out lies inside it)
        if (sb.h_frames) (void)hipHostFree(sb.h_frames);
        if (sb.ev_compute) (void)hipEventDestroy(sb.ev_compute);
        if (sb.ev_done) (void)hipEventDestroy(sb.ev_done);
        if (sb.ev_copy) (void)hipEventDestroy(sb.ev_copy);
    }
    for (auto& e : h->conv_ev) (void)hipEventDestroy(e);
    for (auto& e : h->span_pool) (void)hipEventDestroy(e);
    h->hr.reset(); h->yo.reset(); h->misc.reset(); h->reid.reset();
    if (h->ecc_prev) (void)hipFree(h->ecc_prev);
    if (h->gather_buf) (void)hipFree(h->gather_buf);
    if (h->reid_crops_h) (void)hipHostFree(h->reid_crops_h);
    if (h->reid_feats_h) (void)hipHostFree(h->reid_feats_h);
    if (h->clip_sat_h) (void)hipHostFree(h->clip_sat_h);
    if (h->s_main) (void)hipStreamDestroy(h->s_main);
    if (h->s_det) (void)hipStreamDestroy(h->s_det);
    if (h->s_post) (void)hipStreamDestroy(h->s_post);
    if (h->s_copy) (void)hipStreamDestroy(h->s_copy);
    for (auto st : h->s_br) if (st) (void)hipStreamDestroy(st);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (auto e : h->ev_join) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {h->ev_pre, h->ev_det, h->ev_t0, h->ev_t1}) if (e) (void)hipEventDestroy(e);
    delete h;
}

int eagle_get_config(EagleHandle* h, EagleConfig* cfg)
{
    if (!h || !cfg) return EAGLE_E_INVALID;
    *cfg = h->cfg;
    return EAGLE_OK;
}

const char* eagle_last_error(EagleHandle* h) { return h ? h->err.c_str() : eagle::g_create_error.c_str(); }

int eagle_load_weights(EagleHandle* h, const char* name, const float* data, const int64_t* shape, int ndim)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!name || !data || ndim < 0 || ndim > 8) fail(EAGLE_E_INVALID, "bad weight tensor");
    if (h->finalized) fail(EAGLE_E_STATE, "weights already finalized");
    HostTensor t;
    size_t n = 1;
    if (ndim > 0 && !shape) fail(EAGLE_E_INVALID, "bad weight tensor");
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0 || shape[i] > (1ll << 31) || n * (size_t)shape[i] > ((size_t)1 << 33)) fail(EAGLE_E_INVALID, "%s: bad dimension %lld", name, (long long)shape[i]);
        t.shape.push_back(shape[i]); n *= (size_t)shape[i];
    }
    t.data.assign(data, data + n);
    h->weights[name] = std::move(t);
    API_END(h)
}

int eagle_finalize_weights(EagleHandle* h)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (h->finalized) fail(EAGLE_E_STATE, "already finalized");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    finalize(h);
    API_END(h)
}

int eagle_process_device_frames(EagleHandle* h, const void* d_bgr, int n, EagleFrameResult* out)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!h->finalized) fail(EAGLE_E_STATE, "eagle_finalize_weights has not been called");
    if (!d_bgr || !out || n < 0) fail(EAGLE_E_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    const size_t fsz = (size_t)h->cfg.frame_h * h->cfg.frame_w * 3;
    h->call_steps = (n + h->cfg.batch - 1) / h->cfg.batch;
    const bool direct = !graph_on(h);                             // under graph replay every step goes through the stable staging pointer of its parity (a
    run_pipeline(h, n, out, [&](int p, int i, int na) -> const uint8_t* {   // 2.76-MB-per-frame device copy; the instances are keyed by frame count alone)
        const uint8_t* src = (const uint8_t*)d_bgr + (size_t)i * fsz;
        if (direct) return src;
        HIP_CHECK(hipMemcpyAsync(h->sb[p].d_frames, src, fsz * na, hipMemcpyDeviceToDevice, h->s_main));
        return h->sb[p].d_frames;
    });
    check_saturation(h, "eagle_process_device_frames");
    API_END(h)
}

int eagle_process_frames(EagleHandle* h, const uint8_t* bgr, int n, int64_t frame_stride, int64_t row_stride, EagleFrameResult* out)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!h->finalized) fail(EAGLE_E_STATE, "eagle_finalize_weights has not been called");
    if (!bgr || !out || n < 0) fail(EAGLE_E_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    const int fh = h->cfg.frame_h, fw = h->cfg.frame_w;
    const size_t fsz = (size_t)fh * fw * 3;
    if (row_stride == 0) row_stride = (int64_t)fw * 3;
    if (frame_stride == 0) frame_stride = row_stride * fh;
    // strides in bytes; a row must hold fw BGR pixels and the frames of one call must not overlap (the caller's view may be a crop of a wider
    // surface or a decoder's padded plane: cm.py:568 hands over whatever view it holds).  Negative strides (flipped views) are not supported.
    if (row_stride < (int64_t)fw * 3) fail(EAGLE_E_INVALID, "row_stride %lld is smaller than a row of %d BGR pixels (%d bytes)", (long long)row_stride, fw, fw * 3);
    if (frame_stride < row_stride * (fh - 1) + (int64_t)fw * 3) fail(EAGLE_E_INVALID, "frame_stride %lld is smaller than a frame (%d rows of stride %lld)", (long long)frame_stride, fh, (long long)row_stride);
    // H2D on its own stream: the upload of batch k+1 overlaps the networks of batch k.  (The device staging buffer of parity p
    // was last read by batch k-2, whose records the host has already collected.)
    //   * caller memory that is already pinned (eagle_host_alloc, hipHostMalloc, hipHostRegister): DMA straight out of it;
    //   * pageable caller memory: worker threads copy the batch into a pinned ring slot first — an asynchronous copy out of pageable
    //     memory is staged by the runtime on the calling thread and cost 16 % of the frame rate in round 1.
    hipPointerAttribute_t pa;
    bool pinned = hipPointerGetAttributes(&pa, bgr) == hipSuccess && pa.type == hipMemoryTypeHost;
    (void)hipGetLastError();                               // an unregistered pointer is reported as an error: not one of ours
    if (getenv("EAGLE_H2D_UNSTAGED")) pinned = true;       // developer A/B: round-1 behaviour (pageable memory handed to hipMemcpy2DAsync)
    const bool dense = row_stride == (int64_t)fw * 3 && frame_stride == (int64_t)fsz;
    if (!pinned && !h->pool) {
        int nt = getenv("EAGLE_COPY_THREADS") ? atoi(getenv("EAGLE_COPY_THREADS")) : 8;
        nt = std::max(1, std::min(nt, (int)std::max(1u, std::thread::hardware_concurrency())));
        h->pool.reset(new CopyPool(nt));
    }
    run_pipeline(h, n, out, [&](int p, int i, int na) -> const uint8_t* {
        EagleHandle::StepBuf& sb = h->sb[p];
        hipStream_t sc = h->prof ? h->s_main : h->s_copy;
        const uint8_t* src = bgr + (size_t)i * frame_stride;
        bool src_dense = dense;
        if (!pinned) {
            if (!sb.h_frames) HIP_CHECK(hipHostMalloc((void**)&sb.h_frames, (size_t)h->cfg.batch * fsz, hipHostMallocDefault));
            if (sb.copy_pending) HIP_CHECK(hipEventSynchronize(sb.ev_copy));      // the DMA of batch k-2 has left this slot
            const int parts = 4;                                                  // tasks per frame: keeps every worker busy on small batches
            h->pool->run(na * parts, [&](int t) {
                const int k = t / parts, q = t % parts, r0 = fh * q / parts, r1 = fh * (q + 1) / parts;
                const uint8_t* s0 = src + (size_t)k * frame_stride;
                uint8_t* d0 = sb.h_frames + (size_t)k * fsz;
                if (row_stride == (int64_t)fw * 3) memcpy(d0 + (size_t)r0 * fw * 3, s0 + (size_t)r0 * row_stride, (size_t)(r1 - r0) * fw * 3);
                else for (int r = r0; r < r1; ++r) memcpy(d0 + (size_t)r * fw * 3, s0 + (size_t)r * row_stride, (size_t)fw * 3);
            });
            src = sb.h_frames; src_dense = true;
        }
        if (src_dense) {
            HIP_CHECK(hipMemcpyAsync(sb.d_frames, src, (size_t)na * fsz, hipMemcpyHostToDevice, sc));
        } else {
            for (int k = 0; k < na; ++k)
                HIP_CHECK(hipMemcpy2DAsync(sb.d_frames + (size_t)k * fsz, (size_t)fw * 3, src + (size_t)k * frame_stride,
                                           (size_t)row_stride, (size_t)fw * 3, fh, hipMemcpyHostToDevice, sc));
        }
        if (!h->prof) {
            HIP_CHECK(hipEventRecord(sb.ev_copy, sc));
            sb.copy_pending = true;
            HIP_CHECK(hipStreamWaitEvent(h->s_main, sb.ev_copy, 0));
        }
        return sb.d_frames;
    });
    check_saturation(h, "eagle_process_frames");
    API_END(h)
}

int eagle_reproject(EagleHandle* h, EagleFrameResult* recs, int n, const double* Hs, const uint8_t* flags)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!recs || !Hs || !flags || n < 0) fail(EAGLE_E_INVALID, "bad argument");
    if (n == 0) return EAGLE_OK;
    HIP_CHECK(hipSetDevice(h->cfg.device));
    Net scratch;                                          // owns the three device buffers: freed on every exit path
    EagleFrameResult* d_r = (EagleFrameResult*)scratch.get(sizeof(EagleFrameResult) * (size_t)n);
    double* d_H = (double*)scratch.get(sizeof(double) * 9 * (size_t)n);
    unsigned char* d_f = (unsigned char*)scratch.get((size_t)n);
    HIP_CHECK(hipMemcpyAsync(d_r, recs, sizeof(EagleFrameResult) * (size_t)n, hipMemcpyHostToDevice, h->s_main));
    HIP_CHECK(hipMemcpyAsync(d_H, Hs, sizeof(double) * 9 * (size_t)n, hipMemcpyHostToDevice, h->s_main));
    HIP_CHECK(hipMemcpyAsync(d_f, flags, (size_t)n, hipMemcpyHostToDevice, h->s_main));
    reproject_launch(d_r, d_H, d_f, n, h->cfg.frame_h, h->cfg.frame_w, h->s_main);
    HIP_CHECK(hipMemcpyAsync(recs, d_r, sizeof(EagleFrameResult) * (size_t)n, hipMemcpyDeviceToHost, h->s_main));
    HIP_CHECK(hipStreamSynchronize(h->s_main));
    API_END(h)
}

int eagle_team_colors(EagleHandle* h, const void* d_bgr, int n_frames, const EagleCrop* crops, int n_crops, int32_t* counts)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!d_bgr || n_frames < 0 || n_crops < 0 || (n_crops > 0 && (!crops || !counts))) fail(EAGLE_E_INVALID, "bad argument");
    if (n_crops == 0) return EAGLE_OK;
    HIP_CHECK(hipSetDevice(h->cfg.device));
    Net scratch;
    EagleCrop* d_c = (EagleCrop*)scratch.get(sizeof(EagleCrop) * (size_t)n_crops);
    int* d_n = (int*)scratch.get(sizeof(int) * 12 * (size_t)n_crops);
    HIP_CHECK(hipMemcpyAsync(d_c, crops, sizeof(EagleCrop) * (size_t)n_crops, hipMemcpyHostToDevice, h->s_main));
    eagle::team_colors_launch((const uint8_t*)d_bgr, n_frames, h->cfg.frame_h, h->cfg.frame_w, d_c, n_crops, d_n, h->s_main);
    HIP_CHECK(hipMemcpyAsync(counts, d_n, sizeof(int) * 12 * (size_t)n_crops, hipMemcpyDeviceToHost, h->s_main));
    HIP_CHECK(hipStreamSynchronize(h->s_main));
    API_END(h)
}

int eagle_reid_features(EagleHandle* h, const void* d_bgr, int n_frames, const EagleCrop* crops, int n_crops, float* feats)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!h->finalized || !h->reid) fail(EAGLE_E_STATE, "no appearance network: load the reid.* tensors (OSNet-x0.25, torchreid names) before eagle_finalize_weights");
    if (n_frames < 0 || n_crops < 0 || (n_crops > 0 && (!d_bgr || !crops || !feats))) fail(EAGLE_E_INVALID, "bad argument");
    if (n_crops == 0) return EAGLE_OK;
    HIP_CHECK(hipSetDevice(h->cfg.device));
    struct ProfOff { EagleHandle* h; bool was; ~ProfOff() { h->prof = was; } } prof_guard{h, h->prof};      // restored on every exit path
    h->prof = false;
    for (int i0 = 0; i0 < n_crops; i0 += eagle::REID_NB) {
        const int nb = std::min(eagle::REID_NB, n_crops - i0);
        for (int k = 0; k < eagle::REID_NB; ++k) {
            EagleCrop c = {-1, 0, 0, 0, 0};
            if (k < nb) c = crops[i0 + k];
            h->reid_crops_h[k] = c;
        }
        HIP_CHECK(hipMemcpyAsync(h->reid_crops, h->reid_crops_h, sizeof(EagleCrop) * eagle::REID_NB, hipMemcpyHostToDevice, h->s_main));
        eagle::reid_crop_launch((const uint8_t*)d_bgr, n_frames, h->cfg.frame_h, h->cfg.frame_w, h->reid_crops, eagle::REID_NB, h->reid_in, h->s_main);
        size_t ev_i = 0;
        eagle::run_net(h, h->reid.get(), h->s_main, ev_i);
        HIP_CHECK(hipMemcpyAsync(h->reid_feats_h, h->reid_feats, sizeof(float) * EAGLE_REID_DIM * (size_t)nb, hipMemcpyDeviceToHost, h->s_main));
        HIP_CHECK(hipStreamSynchronize(h->s_main));
        memcpy(feats + (size_t)i0 * EAGLE_REID_DIM, h->reid_feats_h, sizeof(float) * EAGLE_REID_DIM * (size_t)nb);
    }
    API_END(h)
}

int eagle_track_open(EagleHandle* h, const EagleTrackParams* params)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (h->tracker) eagle::tracker_destroy(h->tracker);
    h->tracker = eagle::tracker_create(params);
    h->ecc_has_prev = false;                              // a new BotSort builds a new ECC estimator
    API_END(h)
}

int eagle_track_frames(EagleHandle* h, EagleFrameResult* recs, int n) { return eagle_track_frames_cmc(h, recs, n, nullptr); }

int eagle_track_frames_cmc(EagleHandle* h, EagleFrameResult* recs, int n, const double* warps) { return eagle_track_frames_reid(h, recs, n, warps, nullptr, nullptr, nullptr); }

int eagle_track_frames_reid(EagleHandle* h, EagleFrameResult* recs, int n, const double* warps, const float* feats, const int32_t* feat_det, const int32_t* feat_count)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!recs || n < 0 || (feats && (!feat_det || !feat_count))) fail(EAGLE_E_INVALID, "bad argument");
    if (!h->tracker) fail(EAGLE_E_STATE, "eagle_track_open has not been called");
    if (n == 0) return EAGLE_OK;
    if (feats)                                               // the caller's index arrays are untrusted: validate before anything is dereferenced
        for (int i = 0; i < n; ++i) {
            if (feat_count[i] < 0 || feat_count[i] > EAGLE_MAX_DET) fail(EAGLE_E_INVALID, "eagle_track_frames_reid: feat_count[%d] = %d", i, feat_count[i]);
            if (recs[i].n_det < 0 || recs[i].n_det > EAGLE_MAX_DET) fail(EAGLE_E_INVALID, "eagle_track_frames_reid: record %d has n_det = %d", i, recs[i].n_det);
        }
    if (feats) {
        size_t o = 0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < feat_count[i]; ++k, ++o)
                if (feat_det[o] < 0 || feat_det[o] >= recs[i].n_det)
                    fail(EAGLE_E_INVALID, "eagle_track_frames_reid: feat_det[%zu] = %d is not a detection of record %d (n_det %d)", o, feat_det[o], i, recs[i].n_det);
    }
    std::vector<double> Hs((size_t)n * 9, 0.0);
    std::vector<uint8_t> flags((size_t)n, 0);
    bool any = false;
    size_t fo = 0;                                           // running offset into feats / feat_det
    for (int i = 0; i < n; ++i) {
        const int nf = feats ? feat_count[i] : 0;
        const bool applied = eagle::tracker_apply(h->tracker, recs + i, h->cfg.frame_h, h->cfg.frame_w, h->cfg.detector_conf, warps ? warps + (size_t)i * 6 : nullptr,
                                                  feats ? feats + fo * EAGLE_REID_DIM : nullptr, feats ? feat_det + fo : nullptr, nf);
        fo += (size_t)nf;
        if (!applied) continue;
        any = true;
        flags[i] = recs[i].H_valid ? 1 : 2;                  // re-project the moved foot points with the frame's own homography
        memcpy(&Hs[(size_t)i * 9], recs[i].H, sizeof(double) * 9);
    }
    if (any) {
        const int rc = eagle_reproject(h, recs, n, Hs.data(), flags.data());
        if (rc) return rc;
    }
    API_END(h)
}

#define CLIP_CHECK(h, cond, msg) if (!(h) || !(h)->finalized) return EAGLE_E_STATE; if (!(cond)) { (h)->err = msg; return EAGLE_E_INVALID; }
int eagle_clip_open(EagleHandle* h, const void* d_bgr, int n)
{
    CLIP_CHECK(h, n >= 0 && (d_bgr || n == 0), "eagle_clip_open: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    eagle::clip_open(h, (const uint8_t*)d_bgr, n);
    API_END(h)
}

int eagle_clip_close(EagleHandle* h)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    eagle::clip_close(h);
    API_END(h)
}

int eagle_clip_detect_objects(EagleHandle* h, int first, int count)
{
    CLIP_CHECK(h, h->clip.open && first >= 0 && count >= 0 && first + (int64_t)count <= h->clip.cv.n, "eagle_clip_detect_objects: no open clip or frames out of range")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    eagle::clip_detect_objects(h, first, count);
    API_END(h)
}

int eagle_clip_detect_keypoints(EagleHandle* h, int first, int stride, int count)
{
    CLIP_CHECK(h, h->clip.open && first >= 0 && stride >= 1 && count >= 0 && (count == 0 || first + (int64_t)(count - 1) * stride < h->clip.cv.n),
               "eagle_clip_detect_keypoints: no open clip or frames out of range")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    eagle::clip_detect_keypoints(h, first, stride, count);
    API_END(h)
}

int eagle_clip_get_keypoints(EagleHandle* h, int frame, EagleFlowKp* out, int* n)
{
    CLIP_CHECK(h, h->clip.open && frame >= 0 && frame < h->clip.cv.n && out && n, "eagle_clip_get_keypoints: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    MemList& m = *h->clip.h_mem;                          // mem[] is written on s_main (key-point passes, eagle_clip_set_keypoints)
    HIP_CHECK(hipMemcpyAsync(&m, h->clip.mem + frame, sizeof(m), hipMemcpyDeviceToHost, h->s_main));
    HIP_CHECK(hipStreamSynchronize(h->s_main));
    *n = m.n;
    for (int k = 0; k < m.n && k < EAGLE_N_LANDMARKS; ++k) out[k] = m.kp[k];
    API_END(h)
}

int eagle_clip_set_keypoints(EagleHandle* h, int frame, const EagleFlowKp* in, int n)
{
    CLIP_CHECK(h, h->clip.open && frame >= 0 && frame < h->clip.cv.n && n >= -1 && n <= EAGLE_N_LANDMARKS && (in || n <= 0),
               "eagle_clip_set_keypoints: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    HIP_CHECK(hipStreamSynchronize(h->s_main));           // the staging buffer is free again
    HIP_CHECK(hipStreamSynchronize(h->s_post));           // no loop launch still reads the old entry
    MemList& m = *h->clip.h_mem;
    memset(&m, 0, sizeof(m));
    m.n = n;
    for (int k = 0; k < n; ++k) m.kp[k] = in[k];
    HIP_CHECK(hipMemcpyAsync(h->clip.mem + frame, &m, sizeof(m), hipMemcpyHostToDevice, h->s_main));
    HIP_CHECK(hipEventRecord(h->clip.ev_kp, h->s_main));      // eagle_clip_run orders the loop behind this write
    HIP_CHECK(hipStreamSynchronize(h->s_main));
    API_END(h)
}

int eagle_clip_flow(EagleHandle* h, int src_frame, int dst_frame, int hue_frame, const EagleFlowKp* in, int n_in,
                    EagleFlowKp* out, int* n_out, float* next_pts, uint8_t* status)
{
    CLIP_CHECK(h, h->clip.open && src_frame >= 0 && src_frame < h->clip.cv.n && dst_frame >= 0 && dst_frame < h->clip.cv.n && hue_frame >= 0 &&
               hue_frame < h->clip.cv.n && n_in >= 0 && n_in <= EAGLE_N_LANDMARKS && (in || n_in == 0) && out && n_out,
               "eagle_clip_flow: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    EagleHandle::Clip& c = h->clip;
    *n_out = 0;
    if (n_in == 0) return EAGLE_OK;                       // cm.py:429: empty dict in -> empty dict out
    ChainState& z = *c.h_st;                              // (stream-ordered on s_main behind the gray pyramids; independent of the loop's state)
    memset(&z, 0, sizeof(z));
    z.stalled = -1; z.n_prev = n_in;
    for (int k = 0; k < n_in; ++k) z.prev[k] = in[k];
    HIP_CHECK(hipMemcpyAsync(c.st_op, &z, sizeof(z), hipMemcpyHostToDevice, h->s_main));
    lk_launch(c.cv, src_frame, dst_frame, c.st_op, nullptr, 1, h->s_main);
    flow_filter_launch(c.cv, c.st_op, hue_frame, h->s_main);
    HIP_CHECK(hipMemcpyAsync(&z, c.st_op, sizeof(z), hipMemcpyDeviceToHost, h->s_main));
    HIP_CHECK(hipStreamSynchronize(h->s_main));
    *n_out = z.flow_n;
    for (int k = 0; k < z.flow_n; ++k) out[k] = z.flow[k];
    if (next_pts) memcpy(next_pts, z.lk_next, sizeof(float) * 2 * n_in);
    if (status) memcpy(status, z.lk_status, n_in);
    API_END(h)
}

int eagle_clip_motion(EagleHandle* h, int first, int count, double* warps)
{
    CLIP_CHECK(h, h->clip.open && first >= 0 && count >= 0 && first + count <= h->clip.cv.n && (warps || count == 0), "eagle_clip_motion: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    EagleHandle::Clip& c = h->clip;
    constexpr int GW = 8, GH = 6, NP = GW * GH;              // 48 grid points: one launch of the key-point LK kernel (<= EAGLE_N_LANDMARKS)
    ChainState& z = *c.h_st;
    for (int i = 0; i < count; ++i) {
        double* W = warps + (size_t)i * 6;
        W[0] = 1; W[1] = 0; W[2] = 0; W[3] = 0; W[4] = 1; W[5] = 0;
        const int f = first + i;
        if (f == 0) continue;
        memset(&z, 0, sizeof(z));
        z.stalled = -1; z.n_prev = NP;
        for (int gy = 0; gy < GH; ++gy)
            for (int gx = 0; gx < GW; ++gx) {
                EagleFlowKp& k = z.prev[gy * GW + gx];
                k.label = gy * GW + gx; k.score = 1.f;
                k.x = (int)floor((gx + 0.5) * h->cfg.frame_w / GW); k.y = (int)floor((gy + 0.5) * h->cfg.frame_h / GH);
            }
        HIP_CHECK(hipMemcpyAsync(c.st_op, &z, sizeof(z), hipMemcpyHostToDevice, h->s_main));
        lk_launch(c.cv, f - 1, f, c.st_op, nullptr, 1, h->s_main);
        HIP_CHECK(hipMemcpyAsync(&z, c.st_op, sizeof(z), hipMemcpyDeviceToHost, h->s_main));
        HIP_CHECK(hipStreamSynchronize(h->s_main));
        double p0[2 * NP], p1[2 * NP]; int m = 0;
        for (int k = 0; k < NP; ++k)
            if (z.lk_status[k] == 1) { p0[2 * m] = z.prev[k].x; p0[2 * m + 1] = z.prev[k].y; p1[2 * m] = z.lk_next[2 * k]; p1[2 * m + 1] = z.lk_next[2 * k + 1]; ++m; }
        eagle::similarity_ransac(p0, p1, m, W);
    }
    API_END(h)
}

int eagle_clip_motion_ecc(EagleHandle* h, int first, int count, int carry, double* warps, int* ok_out)
{
    CLIP_CHECK(h, h->clip.open && first >= 0 && count >= 0 && first + count <= h->clip.cv.n && (warps || count == 0), "eagle_clip_motion_ecc: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    EagleHandle::Clip& c = h->clip;
    constexpr double SCALE = 0.15, EPS = 1e-5; constexpr int MAX_ITER = 100;      // boxmot ECC(): scale 0.15, (EPS | COUNT, 100, 1e-5)
    const int dh = (int)lrint(c.cv.h * SCALE), dw = (int)lrint(c.cv.w * SCALE), n = c.cv.n;
    if (dh < 4 || dw < 4) fail(EAGLE_E_INVALID, "eagle_clip_motion_ecc: frame too small for the 0.15-scale alignment");
    hipStream_t sm = h->s_main;
    if (c.ecc_h == 0 && n > 0) {                          // first call of the session (ecc_h is set last: a failed allocation is retried, not half-used)
        if (!c.ecc_small) HIP_CHECK(hipMalloc(&c.ecc_small, (size_t)n * dh * dw));
        if (!c.ecc_pairs) HIP_CHECK(hipMalloc(&c.ecc_pairs, sizeof(int2) * n));
        if (!c.ecc_out) HIP_CHECK(hipMalloc(&c.ecc_out, sizeof(eagle::EccResult) * n));
        HIP_CHECK(hipStreamWaitEvent(sm, c.ev_gray, 0));
        eagle::ecc_small_launch(c.g[0], c.ecc_small, n, c.cv.h, c.cv.w, dh, dw, 1.0 / SCALE, sm);
        c.ecc_h = dh; c.ecc_w = dw;
    }
    const bool use_carry = carry && h->ecc_has_prev && h->ecc_prev_h == dh && h->ecc_prev_w == dw;
    // every adjacent pair at once; pairs behind a failed alignment (boxmot keeps the old template) are re-run one by one below
    std::vector<int2> pairs; std::vector<int> slot(count, -1);
    for (int i = 0; i < count; ++i) {
        const int f = first + i;
        if (f > 0) { slot[i] = (int)pairs.size(); pairs.push_back(make_int2(f - 1, f)); }
        else if (use_carry) { slot[i] = (int)pairs.size(); pairs.push_back(make_int2(-1, f)); }
    }
    std::vector<eagle::EccResult> res(pairs.size());
    if (!pairs.empty()) {
        HIP_CHECK(hipMemcpyAsync(c.ecc_pairs, pairs.data(), sizeof(int2) * pairs.size(), hipMemcpyHostToDevice, sm));
        eagle::ecc_launch(c.ecc_small, h->ecc_prev, c.ecc_pairs, (int)pairs.size(), c.ecc_out, dh, dw, MAX_ITER, EPS, sm);
        HIP_CHECK(hipMemcpyAsync(res.data(), c.ecc_out, sizeof(eagle::EccResult) * pairs.size(), hipMemcpyDeviceToHost, sm));
        HIP_CHECK(hipStreamSynchronize(sm));
    }
    constexpr int NONE = -2;
    int prev = first > 0 ? first - 1 : (use_carry ? -1 : NONE);
    if (first > 0 && c.ecc_next == first && c.ecc_tmpl != NONE && (c.ecc_tmpl >= 0 || use_carry)) prev = c.ecc_tmpl;   // the previous range ended behind a failed alignment
    for (int i = 0; i < count; ++i) {
        const int f = first + i;
        double* W = warps + (size_t)i * 6;
        W[0] = 1; W[1] = 0; W[2] = 0; W[3] = 0; W[4] = 1; W[5] = 0;
        if (ok_out) ok_out[i] = 1;
        if (prev == NONE) { prev = f; continue; }        // the estimator's first frame: identity, becomes the template
        eagle::EccResult r;
        if (slot[i] >= 0 && pairs[slot[i]].x == prev) r = res[slot[i]];
        else {
            const int2 one = make_int2(prev, f);
            HIP_CHECK(hipMemcpyAsync(c.ecc_pairs, &one, sizeof(one), hipMemcpyHostToDevice, sm));
            eagle::ecc_launch(c.ecc_small, h->ecc_prev, c.ecc_pairs, 1, c.ecc_out, dh, dw, MAX_ITER, EPS, sm);
            HIP_CHECK(hipMemcpyAsync(&r, c.ecc_out, sizeof(r), hipMemcpyDeviceToHost, sm));
            HIP_CHECK(hipStreamSynchronize(sm));
        }
        if (!r.ok) { if (ok_out) ok_out[i] = 0; continue; }     // cv2 raised: identity, template unchanged
        for (int k = 0; k < 6; ++k) W[k] = (double)r.M[k];
        W[2] = (double)(float)((double)r.M[2] / SCALE); W[5] = (double)(float)((double)r.M[5] / SCALE);   // warp_matrix[i, 2] /= self.scale
        prev = f;
    }
    if (count > 0) { c.ecc_next = first + count; c.ecc_tmpl = prev; }
    if (carry && prev != NONE && prev != -1) {
        if (!h->ecc_prev || h->ecc_prev_h != dh || h->ecc_prev_w != dw) {
            if (h->ecc_prev) HIP_CHECK(hipFree(h->ecc_prev));
            h->ecc_prev = nullptr;
            HIP_CHECK(hipMalloc(&h->ecc_prev, (size_t)dh * dw));
            h->ecc_prev_h = dh; h->ecc_prev_w = dw;
        }
        HIP_CHECK(hipMemcpyAsync(h->ecc_prev, c.ecc_small + (size_t)prev * dh * dw, (size_t)dh * dw, hipMemcpyDeviceToDevice, sm));
        HIP_CHECK(hipStreamSynchronize(sm));
        h->ecc_has_prev = true;
    }
    API_END(h)
}

int eagle_clip_run(EagleHandle* h, int first, int last, int keypoint_interval, int homography_interval, int calibration, int wait, int* stalled_at)
{
    CLIP_CHECK(h, h->clip.open && first >= 0 && first <= last && last <= h->clip.cv.n && keypoint_interval >= 1 && homography_interval >= 1,
               "eagle_clip_run: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    EagleHandle::Clip& c = h->clip;
    hipStream_t sp = h->s_post;
    HIP_CHECK(hipStreamWaitEvent(sp, c.ev_gray, 0));
    HIP_CHECK(hipStreamWaitEvent(sp, c.ev_det, 0));       // every detector / HRNet pass enqueued so far
    HIP_CHECK(hipStreamWaitEvent(sp, c.ev_kp, 0));
    if (first < last) {
        if (first == 0) HIP_CHECK(hipMemcpyAsync(c.st, c.h_zero, sizeof(ChainState), hipMemcpyHostToDevice, sp));
        else if (wait) HIP_CHECK(hipMemcpyAsync((char*)c.st + offsetof(ChainState, stalled), &c.h_tail[2], sizeof(int), hipMemcpyHostToDevice, sp));   // resume
        // (an asynchronous call for a later chunk must NOT clear the flag: if an earlier chunk stalled, its launches have to fall through too)
    }
    for (int i = first; i < last; ++i) {
        lk_launch(c.cv, i > 0 ? i - 1 : 0, i, c.st, c.mem, keypoint_interval, sp);
        chain_launch(c.cv, c.st, c.mem, c.recs, h->pp, i, keypoint_interval, homography_interval, calibration, sp);
    }
    HIP_CHECK(hipEventRecord(c.ev_loop, sp));
    if (stalled_at) *stalled_at = -1;
    if (wait) {
        HIP_CHECK(hipMemcpyAsync(c.h_tail, (char*)c.st + offsetof(ChainState, stalled), sizeof(int) * 2, hipMemcpyDeviceToHost, sp));
        HIP_CHECK(hipStreamSynchronize(sp));
        if (stalled_at) *stalled_at = c.h_tail[0];
        if (c.h_tail[1]) { h->err = "the reference raises IndexError in calibrate_keypoints at frame " + std::to_string(c.h_tail[1] - 1); return EAGLE_E_REFERENCE_RAISES; }
    }
    API_END(h)
}

int eagle_clip_fetch(EagleHandle* h, EagleFrameResult* out)
{
    CLIP_CHECK(h, h->clip.open && (out || h->clip.cv.n == 0), "eagle_clip_fetch: bad arguments")
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    eagle::clip_sync(h);
    if (h->clip.cv.n > 0) HIP_CHECK(hipMemcpy(out, h->clip.recs, sizeof(EagleFrameResult) * (size_t)h->clip.cv.n, hipMemcpyDeviceToHost));
    // saturated f32s stores of the session's detector / key-point passes (counted per slot of the device batch, not per clip frame)
    HIP_CHECK(hipMemcpy(h->clip_sat_h, h->clip_sat, eagle::sat_pad_bytes(h->cfg.batch), hipMemcpyDeviceToHost));
    h->sat_events = 0; h->sat_frames = 0;
    for (int i = 0; i < h->cfg.batch; ++i) if (h->clip_sat_h[i]) { h->sat_events += h->clip_sat_h[i]; ++h->sat_frames; }
    h->timings.sat_events = (int32_t)std::min<long long>(h->sat_events, 0x7fffffff); h->timings.sat_frames = h->sat_frames;
    eagle::check_saturation(h, "eagle_clip_fetch");
    API_END(h)
}

int eagle_device_alloc(EagleHandle* h, int64_t bytes, void** dptr)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    HIP_CHECK(hipSetDevice(h->cfg.device));
    HIP_CHECK(hipMalloc(dptr, (size_t)bytes));
    API_END(h)
}
int eagle_device_free(EagleHandle* h, void* dptr)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    HIP_CHECK(hipFree(dptr));
    API_END(h)
}
int eagle_device_upload(EagleHandle* h, void* dptr, const void* src, int64_t bytes)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    HIP_CHECK(hipMemcpy(dptr, src, (size_t)bytes, hipMemcpyHostToDevice));
    API_END(h)
}

int eagle_host_alloc(EagleHandle* h, int64_t bytes, void** ptr)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!ptr || bytes < 0) fail(EAGLE_E_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    HIP_CHECK(hipHostMalloc(ptr, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocDefault));
    API_END(h)
}
int eagle_host_free(EagleHandle* h, void* ptr)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    HIP_CHECK(hipHostFree(ptr));
    API_END(h)
}

int eagle_set_profiling(EagleHandle* h, int on)
{
    if (!h) return EAGLE_E_INVALID;
    h->prof = on != 0;
    if (h->prof) h->ktab.clear();
    return EAGLE_OK;
}
int eagle_get_kernel_times(EagleHandle* h, EagleKernelTime* out, int cap, int* n)
{
    if (!h || !n || (cap > 0 && !out)) return EAGLE_E_INVALID;
    *n = (int)h->ktab.size();
    for (int i = 0; i < *n && i < cap; ++i) out[i] = h->ktab[i];
    return EAGLE_OK;
}
int eagle_get_timings(EagleHandle* h, EagleTimings* t)
{
    if (!h || !t) return EAGLE_E_INVALID;
    *t = h->timings;
    t->graph_captures = h->graph_captures; t->graph_skipped = h->graph_skipped;
    return EAGLE_OK;
}

int eagle_debug(const char* key, int64_t value, void* out, int64_t out_bytes)
{
    EagleHandle* h = nullptr;
    int rc = 0;
    API_BEGIN
    // The switches are process-wide and change what every handle computes ("skip" drops whole kernels): they only exist for the developer
    // probes under tools/ and stay inert unless the process opts in.
    static const bool enabled = getenv("EAGLE_ENABLE_DEBUG") && atoi(getenv("EAGLE_ENABLE_DEBUG")) != 0;
    if (!enabled) fail(EAGLE_E_STATE, "eagle_debug is disabled: set EAGLE_ENABLE_DEBUG=1 in the environment of a developer probe to use it");
    if (key && !strcmp(key, "skip")) { eagle::g_dbg_skip = (int)value; return EAGLE_OK; }
    rc = eagle::lk_debug(key, value, out, out_bytes);
    if (rc) return EAGLE_E_INVALID;
    API_END(h)
}

// ---- RCCL gather (resolved lazily with dlopen so the library loads on hosts without RCCL) ------------------------
typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_uid)(nccl_uid*);
typedef int (*fn_init)(void**, int, nccl_uid, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

// A process must hold ONE copy of the ROCm runtime stack.  PyTorch-ROCm wheels bundle their own (torch/lib/libamdhip64.so, libhsa-runtime64.so,
// librccl.so — SONAMEs libamdhip64.so.7 / librccl.so.1, the same as /opt/rocm's), so in a process that has imported torch the RCCL to use is the
// one that is already mapped: RTLD_NOLOAD by SONAME finds it (round 3 opened "librccl.so.1" RTLD_GLOBAL from the search path, which next to an
// already-imported torch could map a second RCCL against a second HIP runtime — DESIGN.md §8).  Only a torch-free process loads /opt/rocm's copy,
// and never RTLD_GLOBAL: nothing else resolves symbols through it.
static void* rccl_lib()
{
    static void* lib = nullptr;
    if (!lib) {
        for (const char* n : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (lib) return lib;
        }
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
    }
    return lib;
}

int eagle_comm_id(void* id128)
{
    EagleHandle* h = nullptr;
    API_BEGIN
    void* lib = rccl_lib();
    if (!lib) fail(EAGLE_E_COMM, "librccl not found: %s", dlerror());
    fn_uid f = (fn_uid)dlsym(lib, "ncclGetUniqueId");
    if (!f) fail(EAGLE_E_COMM, "ncclGetUniqueId missing");
    nccl_uid id;
    const int rc = f(&id);
    if (rc) fail(EAGLE_E_COMM, "ncclGetUniqueId failed: %d", rc);
    memcpy(id128, &id, 128);
    API_END(h)
}

int eagle_comm_init(EagleHandle* h, int rank, int world, const void* id128)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    void* lib = rccl_lib();
    if (!lib) fail(EAGLE_E_COMM, "librccl not found");
    fn_init f = (fn_init)dlsym(lib, "ncclCommInitRank");
    if (!f) fail(EAGLE_E_COMM, "ncclCommInitRank missing");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    nccl_uid id;
    memcpy(&id, id128, 128);
    const int rc = f(&h->comm, world, id, rank);
    if (rc) fail(EAGLE_E_COMM, "ncclCommInitRank failed: %d", rc);
    h->rank = rank; h->world = world; h->rccl = lib;
    API_END(h)
}

int eagle_gather(EagleHandle* h, const EagleFrameResult* local, int n_local, EagleFrameResult* all)
{
    if (!h) return EAGLE_E_INVALID;
    API_BEGIN
    if (!local || !all || n_local < 0) fail(EAGLE_E_INVALID, "bad argument");
    const size_t bytes = sizeof(EagleFrameResult) * (size_t)n_local;
    if (h->world == 1 && !h->comm) { memcpy(all, local, bytes); return EAGLE_OK; }
    if (!h->comm) fail(EAGLE_E_STATE, "eagle_comm_init has not been called");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    fn_allgather ag = (fn_allgather)dlsym(h->rccl, "ncclAllGather");
    if (!ag) fail(EAGLE_E_COMM, "ncclAllGather missing");
    // device staging of the collective: kept with the handle and only ever grown (a hipMalloc / hipFree pair per call sat inside the bench's timed region)
    const size_t need = std::max<size_t>(bytes * (size_t)(h->world + 1), 256);
    if (h->gather_cap < need) {
        if (h->gather_buf) { (void)hipFree(h->gather_buf); h->gather_buf = nullptr; h->gather_cap = 0; }
        HIP_CHECK(hipMalloc(&h->gather_buf, need));
        h->gather_cap = need;
    }
    void *d_send = h->gather_buf, *d_recv = (char*)h->gather_buf + bytes;
    HIP_CHECK(hipMemcpyAsync(d_send, local, bytes, hipMemcpyHostToDevice, h->s_main));
    const int rc = ag(d_send, d_recv, bytes, /*ncclChar*/ 0, h->comm, h->s_main);
    if (rc) fail(EAGLE_E_COMM, "ncclAllGather failed: %d", rc);
    HIP_CHECK(hipMemcpyAsync(all, d_recv, bytes * h->world, hipMemcpyDeviceToHost, h->s_main));
    HIP_CHECK(hipStreamSynchronize(h->s_main));
    API_END(h)
}

// ---- operator-level entry points for the parity tests ---------------------------------------------------------------
static void to_dev(Net& net, int prec, const float* src, int n, int h, int w, int c, int cpad, TView& v)
{
    v.n = n; v.h = h; v.w = w; v.c = cpad; v.cs = cpad; v.off = 0; v.f32 = prec_tensor_fmt(prec);
    const size_t px = (size_t)n * h * w;
    if (v.f32 == 2) {                                       // [hi x 8][lo x 8] per 8 channels, hi = rn(16 v), lo = rn(16 v - hi)
        std::vector<_Float16> t(px * cpad * 2, (_Float16)0.f);
        for (size_t p = 0; p < px; ++p)
            for (int k = 0; k < c; ++k) {
                const float sv = src[p * c + k] * 16.0f;
                const _Float16 hi = (_Float16)sv;
                t[p * cpad * 2 + (k / 8) * 16 + (k % 8)] = hi;
                t[p * cpad * 2 + (k / 8) * 16 + 8 + (k % 8)] = (_Float16)(sv - (float)hi);
            }
        v.p = net.upload(t.data(), t.size() * 2);
    } else if (v.f32) {
        std::vector<float> t(px * cpad, 0.f);
        for (size_t p = 0; p < px; ++p) for (int k = 0; k < c; ++k) t[p * cpad + k] = src[p * c + k];
        v.p = net.upload(t.data(), t.size() * 4);
    } else {
        std::vector<_Float16> t(px * cpad, (_Float16)0.f);
        for (size_t p = 0; p < px; ++p) for (int k = 0; k < c; ++k) t[p * cpad + k] = (_Float16)src[p * c + k];
        v.p = net.upload(t.data(), t.size() * 2);
    }
}
static void from_dev(const TView& v, int c, float* dst)
{
    const size_t px = (size_t)v.n * v.h * v.w;
    if (v.f32 == 2) {
        std::vector<_Float16> t(px * v.cs * 2);
        HIP_CHECK(hipMemcpy(t.data(), v.p, t.size() * 2, hipMemcpyDeviceToHost));
        for (size_t p = 0; p < px; ++p)
            for (int k = 0; k < c; ++k) {
                const size_t e = p * v.cs * 2 + (size_t)((v.off + k) / 8) * 16 + (v.off + k) % 8;
                dst[p * c + k] = ((float)t[e] + (float)t[e + 8]) * 0.0625f;
            }
    } else if (v.f32) {
        std::vector<float> t(px * v.cs);
        HIP_CHECK(hipMemcpy(t.data(), v.p, t.size() * 4, hipMemcpyDeviceToHost));
        for (size_t p = 0; p < px; ++p) for (int k = 0; k < c; ++k) dst[p * c + k] = t[p * v.cs + v.off + k];
    } else {
        std::vector<_Float16> t(px * v.cs);
        HIP_CHECK(hipMemcpy(t.data(), v.p, t.size() * 2, hipMemcpyDeviceToHost));
        for (size_t p = 0; p < px; ++p) for (int k = 0; k < c; ++k) dst[p * c + k] = (float)t[p * v.cs + v.off + k];
    }
}

int eagle_op_conv2d(int device, int precision, const float* x, int n, int h, int w, int cin, const float* w_hwio,
                    const float* bias, int cout, int ks, int stride, int pre_act, const float* r1, const float* r2,
                    int post_act, float* y)
{
    EagleHandle* hh = nullptr;
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    Net net;
    const int g = precision == EAGLE_PREC_F32 ? 4 : 8;
    const int cin_pad = cin <= g ? g : (cin + 15) / 16 * 16, cout_pad = (cout + 15) / 16 * 16;
    const int ho = (h + 2 * (ks / 2) - ks) / stride + 1, wo = (w + 2 * (ks / 2) - ks) / stride + 1;
    ConvLaunch L;
    to_dev(net, precision, x, n, h, w, cin, cin_pad, L.x);
    L.cfg = conv_choose(precision, ks, stride, cin_pad, cout_pad, wo, pre_act == ACT_NONE && post_act <= ACT_RELU, r1 && r2, r1 || r2);
    if (!conv_supported(precision, L.cfg)) fail(EAGLE_E_NOKERNEL, "no kernel instance ks=%d s=%d kc=%d nt=%d", ks, stride, L.cfg.kc, L.cfg.nt);
    std::vector<char> tiled(conv_weight_elems(precision, L.cfg) * (precision == EAGLE_PREC_F32 ? 4 : 2));
    conv_tile_weights(precision, L.cfg, w_hwio, cin, cout, tiled.data(), &L.descale);
    L.w = net.upload(tiled.data(), tiled.size());
    std::vector<float> b(cout_pad, 0.f);
    for (int i = 0; i < cout; ++i) b[i] = bias[i];
    L.bias = (const float*)net.upload(b.data(), b.size() * 4);
    L.y.n = n; L.y.h = ho; L.y.w = wo; L.y.c = cout_pad; L.y.cs = cout_pad; L.y.f32 = prec_tensor_fmt(precision);
    L.y.p = net.get((size_t)n * ho * wo * cout_pad * L.y.esize());
    if (r1) to_dev(net, precision, r1, n, ho, wo, cout, cout_pad, L.r1);
    if (r2) to_dev(net, precision, r2, n, ho, wo, cout, cout_pad, L.r2);
    L.pre_act = pre_act; L.post_act = post_act;
    conv_launch(precision, L, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    from_dev(L.y, cout, y);
    API_END(hh)
}

int eagle_op_bottleneck(int device, const float* x, int n, int h, int w, int cin, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* w3, const float* b3, const float* res, float* y, int reps, float* ms, const float* wd, const float* bd)
{
    EagleHandle* hh = nullptr;
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    if (!x || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !y || n < 1 || h < 1 || w < 1 || cin % 16 || cin < 16) fail(EAGLE_E_INVALID, "eagle_op_bottleneck: bad argument (Cin must be a multiple of 16)");
    if ((wd != nullptr) != (bd != nullptr) || (wd && (res || cin != 64))) fail(EAGLE_E_INVALID, "eagle_op_bottleneck: the in-kernel downsample branch takes (wd, bd) together, Cin = 64 and no residual tensor");
    if (!res && !wd && cin != 256) fail(EAGLE_E_INVALID, "eagle_op_bottleneck: an identity shortcut needs Cin = 256");
    Net net;
    BneckLaunch L;
    to_dev(net, EAGLE_PREC_F32S, x, n, h, w, cin, cin, L.x);
    if (res) to_dev(net, EAGLE_PREC_F32S, res, n, h, w, 256, 256, L.res); else L.res = L.x;
    L.y = L.x; L.y.c = L.y.cs = 256; L.y.off = 0; L.y.p = net.get((size_t)n * h * w * 256 * 4);
    if (wd) L.res = L.y;                                    // (not read)
    std::vector<_Float16> img;
    bneck_tile_weights(w1, 1, cin, 64, img, &L.ds1); L.w1 = net.upload(img.data(), img.size() * 2);
    bneck_tile_weights(w2, 9, 64, 64, img, &L.ds2); L.w2 = net.upload(img.data(), img.size() * 2);
    std::vector<float> w3x(w3, w3 + 64 * 256), sb3(b3, b3 + 256);
    if (wd) {                                               // K = 128: [W3 | Wd], bias b3 + bd
        w3x.insert(w3x.end(), wd, wd + 64 * 256);
        for (int o = 0; o < 256; ++o) sb3[o] = sb3[o] + bd[o];
        L.ds_fused = true;
    }
    bneck_tile_weights(w3x.data(), 1, wd ? 128 : 64, 256, img, &L.ds3); L.w3 = net.upload(img.data(), img.size() * 2);
    std::vector<float> sb1(b1, b1 + 64), sb2(b2, b2 + 64);
    bneck_scale_bias(sb1, L.ds1); bneck_scale_bias(sb2, L.ds2); bneck_scale_bias(sb3, L.ds3);
    L.b1 = (const float*)net.upload(sb1.data(), 64 * 4); L.b2 = (const float*)net.upload(sb2.data(), 64 * 4); L.b3 = (const float*)net.upload(sb3.data(), 256 * 4);
    if (getenv("EAGLE_BNECK_TIMING")) L.dbg = (unsigned long long*)net.get(8192 * 8 * 8);      // (developer timing builds: -DEAGLE_BNECK_TIMING)
    unsigned* op_sat = nullptr; unsigned* const* op_sat_slot = &op_sat;
    if (getenv("EAGLE_BNECK_OPSAT")) { op_sat = (unsigned*)net.get(sizeof(unsigned) * (size_t)n); L.sat_slot = op_sat_slot; }      // developer: the per-frame saturation counters the pipeline passes
    bneck_launch(L, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (L.dbg) {
        std::vector<unsigned long long> t(8192 * 8);
        HIP_CHECK(hipMemcpy(t.data(), L.dbg, t.size() * 8, hipMemcpyDeviceToHost));
        double sum[8] = {0}; int nw = 0;
        for (int b = 0; b < 8192; ++b) { bool any = false; for (int k = 0; k < 8; ++k) { sum[k] += (double)t[b * 8 + k]; any |= t[b * 8 + k] != 0; } nw += any; }
        if (nw) fprintf(stderr, "[bneck timing] %d workgroups, mean us per workgroup: phase1 %.1f  wait %.1f  epi1 %.1f  phase2 %.1f  epi2 %.1f  phase3 %.1f\n", nw,
                        sum[0] / nw / 100, sum[1] / nw / 100, sum[2] / nw / 100, sum[3] / nw / 100, sum[4] / nw / 100, sum[5] / nw / 100);
    }
    if (reps > 0 && ms) {                                   // developer timing: the launch alone, HIP events on the launch stream
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) bneck_launch(L, nullptr);
        HIP_CHECK(hipEventRecord(e1, nullptr));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(ms, e0, e1));
        *ms /= (float)reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    from_dev(L.y, 256, y);
    API_END(hh)
}

int eagle_op_fuse_sum(int device, int precision, const float* base, int n, int H, int W, int c, int n_up,
                      const float* const* ups, const int* up_h, const int* up_w, int relu, float* y)
{
    EagleHandle* hh = nullptr;
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    if (n_up > 3 || c % 8) fail(EAGLE_E_INVALID, "fuse_sum: n_up <= 3 and c %% 8 == 0 required");
    Net net;
    TView b, o;
    to_dev(net, precision, base, n, H, W, c, c, b);
    FuseUp u[3];
    for (int i = 0; i < n_up; ++i) to_dev(net, precision, ups[i], n, up_h[i], up_w[i], c, c, u[i].z);
    o = b; o.p = net.get((size_t)n * H * W * c * b.esize());
    fuse_sum_launch(b, u, n_up, relu, o, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    from_dev(o, c, y);
    API_END(hh)
}

int eagle_op_preprocess(int device, int precision, const uint8_t* bgr, int n, int h, int w, int det_imgsz,
                        float* kp_out, float* det_out, int* det_hw)
{
    return eagle_op_preprocess_lb(device, precision, bgr, n, h, w, det_imgsz, EAGLE_LETTERBOX_RECT, kp_out, det_out, det_hw);
}

int eagle_op_preprocess_lb(int device, int precision, const uint8_t* bgr, int n, int h, int w, int det_imgsz, int letterbox,
                           float* kp_out, float* det_out, int* det_hw)
{
    EagleHandle* hh = nullptr;
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    Net net;
    const int cp = precision == EAGLE_PREC_F32 ? 4 : 8;
    const LetterBox lb = letterbox_geometry(h, w, det_imgsz, letterbox);
    det_hw[0] = lb.out_h; det_hw[1] = lb.out_w;
    if (!kp_out || !det_out) return EAGLE_OK;
    uint8_t* d = (uint8_t*)net.upload(bgr, (size_t)n * h * w * 3);
    TView kp, det;
    kp.n = n; kp.h = 540; kp.w = 960; kp.c = kp.cs = cp; kp.f32 = prec_tensor_fmt(precision);
    det = kp; det.h = lb.out_h; det.w = lb.out_w;
    kp.p = net.get((size_t)n * 540 * 960 * cp * kp.esize());
    det.p = net.get((size_t)n * lb.out_h * lb.out_w * cp * det.esize());
    preprocess_launch(precision, d, n, h, w, kp, det, lb, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    from_dev(kp, 3, kp_out);
    from_dev(det, 3, det_out);
    API_END(hh)
}

int eagle_op_find_homography(int device, const float* img_pts, const float* world_pts, int n, double thresh,
                             int max_iters, int lm_iters, double* H9, uint8_t* mask, int* ok)
{
    EagleHandle* hh = nullptr;
    API_BEGIN
    HIP_CHECK(hipSetDevice(device));
    if (n < 0 || n > EAGLE_MAX_KP) fail(EAGLE_E_INVALID, "0 <= n <= %d required", EAGLE_MAX_KP);
    Net net;
    float* di = (float*)net.upload(img_pts, sizeof(float) * 2 * std::max(n, 1));
    float* dw = (float*)net.upload(world_pts, sizeof(float) * 2 * std::max(n, 1));
    double* dH = (double*)net.get(72);
    uint8_t* dm = (uint8_t*)net.get(256);
    int* dok = (int*)net.get(16);
    homography_only_launch(di, dw, n, thresh, max_iters, lm_iters, dH, dm, dok, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(H9, dH, 72, hipMemcpyDeviceToHost));
    if (n > 0) HIP_CHECK(hipMemcpy(mask, dm, n, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(ok, dok, sizeof(int), hipMemcpyDeviceToHost));
    API_END(hh)
}

}  // extern "C"
