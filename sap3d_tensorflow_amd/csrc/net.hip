// Host runtime of libp3dhip: builds the P3D graph once (static shapes, everything resident in
// HBM), then runs forward / backward / Adam as a fixed list of kernel launches on one HIP stream,
// with RCCL gradient all-reduce on a side stream.  Graph structure follows the reference's
// graph-building functions (p3d.py:10-221) but nothing else of TF's runtime is mirrored: there is
// no session, no tracing, no host-resident variables.
//
// Parameters live in ONE flat fp32 buffer in creation (= forward) order with matching flat
// gradient / Adam-moment buffers, so the optimiser is one kernel and gradient buckets for the
// all-reduce are contiguous ranges that complete back-to-front during backward.
#include "../../include/p3d_hip.h"
#include "p3d_kernels.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

// The product's own runtime switches (everything else is a -DP3D_TUNING switch, p3d_tune_env): read once.
struct RuntimeEnv { bool debug_sync = false, graph = false, no_side_stream = false; };
const RuntimeEnv& runtime_env() {
    static const RuntimeEnv env = [] {
        RuntimeEnv e;
        e.debug_sync = getenv("P3D_DEBUG_SYNC") != nullptr;                                   // synchronise and log after every op
        if (const char* v = getenv("P3D_GRAPH")) e.graph = atoi(v) != 0 && !e.debug_sync;      // captured step graph (slower on ROCm 7.2)
        e.no_side_stream = getenv("P3D_NO_SIDE_STREAM") != nullptr;                            // everything on one stream
        return e;
    }();
    return env;
}
// all-reduce bucket size in MB (also fixes where queued filter gradients are flushed); read whenever a handle or a communicator
// is created, so that one process can build handles with different bucket sizes (tests/test_gpu_dp.py); 0: the default
long bucket_mb_env() { const char* e = getenv("P3D_BUCKET_MB"); return e ? atol(e) : 0; }

struct P3dError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIPCHECK(expr)                                                                               \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            throw P3dError(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                           std::to_string(__LINE__) + ")");                                          \
    } while (0)
#define NCCLCHECK(expr)                                                                              \
    do {                                                                                             \
        ncclResult_t r_ = (expr);                                                                    \
        if (r_ != ncclSuccess) throw P3dError(std::string(#expr) + " failed: " + ncclGetErrorString(r_)); \
    } while (0)

enum InitKind { INIT_XAVIER = 0, INIT_ZEROS = 1, INIT_ONES = 2, INIT_VS = 3 };

struct Param {
    std::string name;
    std::vector<int64_t> shape;
    int64_t count = 0, off = 0;
    bool trainable = true;
    int init = INIT_XAVIER;
    float* p = nullptr;     // device
    float* g = nullptr;     // device gradient (trainables)
};

// The stem: conv -> BatchNorm -> ReLU, and the conv has no input gradient.  The normalisation's backward then ends after its
// reduce / finalize launches and hands its arguments to the conv's filter gradient, which applies it on its operand path.
struct StemBnLink { bool enabled = false, pending = false; BnBwdArgs args; };

struct Act {
    std::string name;
    int N = 0, D = 0, H = 0, W = 0, C = 0, ld = 0;
    float* p = nullptr;
    float* g = nullptr;
    Act* parent = nullptr;          // channel-slice view of parent's storage
    std::vector<Act*> views;
    char* last_flag = nullptr;      // accumulate-flag of the most recently registered consumer
    bool whole_consumed = false;    // a consumer of the whole buffer (all views) is registered
    // BatchNorm fusion: a normalised tensor that the fused forward never stores; reading it (p3d_get_activation) runs this
    std::function<void(hipStream_t)> materialize;
    // the stem conv's output: its BatchNorm's backward may leave the apply pass to the conv's filter gradient (stem_wgrad.hip)
    std::shared_ptr<struct StemBnLink> stem_link;
    int64_t rows() const { return (int64_t)N * D * H * W; }
};

struct BN {
    std::string name;
    int C = 0;
    Param *gamma = nullptr, *beta = nullptr, *mm = nullptr, *mv = nullptr;
    int64_t part_off = -1;          // statistics partials [part_cap][C][2] floats in statpart_arena (-1: producer has no epilogue)
    int part_cap = 0;
    int nparts = 0;                 // partials the last forward's producer wrote (host-side, set at enqueue)
    float *scale = nullptr, *shift = nullptr, *mean = nullptr, *invstd = nullptr;
    bool follows_flag = false;      // obeys the `training` placeholder (stem / decoder); else always batch stats
    bool used_batch = true;         // what the last forward used
    // BatchNorm fusion (p3d_kernels.h): backward partials (sum g, sum g*xhat) per output tile of the gating launch, and the
    // published k1 / k2 / k3 of dy = k1*g + k2*y + k3
    bool fusable = false;
    int64_t gpart_off = -1; int gpart_cap = 0; int gnparts = 0;
    float* coef = nullptr;
};

// Per-launch HIP-event timing (p3d_profile_step): one record per kernel launch, on the launch stream.
struct ProfRec {
    std::string kernel, op;
    double flops = 0, bytes = 0;
    int phase = 0;                 // 0 forward, 1 backward, 2 optimiser
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
struct Prof {
    std::vector<ProfRec> recs;
    int phase = 0;
    std::string cur_op;
};

struct GN {                            // one GroupNorm layer (gn/p3d_gn.py:24-46)
    std::string name;
    int C = 0, G = 0, N = 0;
    Param *gamma = nullptr, *beta = nullptr;
    int64_t sums_off = 0;              // forward (sum, sumsq) in the stats arena   [N][C][2] doubles
    int64_t bsums_off = 0;             // backward sums in the reduction arena       [N][C][2] doubles
    int64_t tab_off = 0;               // scale, shift, mean, invstd [N][C] each + coef [N][C][3], in bnbuf
};

struct CbamSite {                      // one cbam_block on a bottleneck residual (utils/network.py:198-274)
    Param *k0 = nullptr, *b0 = nullptr, *k1 = nullptr, *b1 = nullptr, *k7 = nullptr;
    Act* x = nullptr;
    float* dout = nullptr;             // gradient of the CBAM output, written by the block-end pass
    int chunks = 1;
    int64_t buf_off = 0;               // float scratch in bnbuf
    char* xflag = nullptr;
};

struct Ctx {
    bool training = false;
    float drop = 0.f;
    uint64_t seed = 0;
    const unsigned long long* seed_dev = nullptr;   // captured step graphs: dropout seed and Adam step size live in device memory
    const float* lr_dev = nullptr;
    bool update_moving = false;
    bool per_sample = false;          // batch-statistics BNs normalise every clip by its own statistics (p3d_predict_windows)
    bool fuse = false;                // BatchNorm fused into the neighbouring convs' operand paths (set by run_forward / run_backward)
    bool fuse_bwd = false;            // ... in the backward pass as well (else: BatchNorm's backward keeps its own launches)
    hipStream_t s = nullptr;
    Prof* prof = nullptr;
    hipStream_t side = nullptr;       // weight gradients run here, off the backward critical path
    // planning pass (finalize_build): nothing is launched, zero-fill requests are recorded instead
    std::vector<std::pair<float*, size_t>>* dry = nullptr;
    // buffers inside [z0, z1) are zeroed wholesale at the start of the phase: no per-op memset
    const char* z0 = nullptr; const char* z1 = nullptr;
    // backward of the decoder: side-stream jobs are parked here and released when the walk reaches the encoder (run_backward)
    std::vector<std::pair<hipEvent_t, std::function<void(const Ctx&)>>>* defer = nullptr;
    const char* bwd_op = nullptr;     // name of the op whose backward is running (diagnostics)
};

// ---- schedule trace (test hook p3d_debug_schedule): every stream operation of a pass -- kernel launches, async fills, event
//      records and waits, all-reduce launches -- in host issue order, with streams and events named by small ids.  The ordering
//      bugs this file can have (a fill that is not ordered against the launch behind it, a hand-over before its producer) do
//      not show as numbers until another session reuses the state; they do show in who waits for whom.
struct SchedTrace {
    std::vector<std::string> lines;
    std::map<hipStream_t, std::string> streams;
    std::map<hipEvent_t, int> events;
    std::string sname(hipStream_t s) {
        auto it = streams.find(s);
        if (it != streams.end()) return it->second;
        const std::string n = "s" + std::to_string(streams.size());
        streams[s] = n;
        return n;
    }
    int eid(hipEvent_t e) {
        auto it = events.find(e);
        if (it != events.end()) return it->second;
        const int n = (int)events.size();
        events[e] = n;
        return n;
    }
};
thread_local SchedTrace* g_trace = nullptr;
inline hipError_t ev_record(hipEvent_t e, hipStream_t s) {
    if (g_trace) g_trace->lines.push_back("R " + g_trace->sname(s) + " e" + std::to_string(g_trace->eid(e)));
    return hipEventRecord(e, s);
}
inline hipError_t ev_wait(hipStream_t s, hipEvent_t e) {
    if (g_trace) g_trace->lines.push_back("W " + g_trace->sname(s) + " e" + std::to_string(g_trace->eid(e)));
    return hipStreamWaitEvent(s, e, 0);
}
// Every fill and copy of this library NAMES ITS STREAM (round 5).  A bare hipMemset / hipMemcpy runs on the null stream, which
// is not ordered against the handle's non-blocking streams: twice in two rounds such a call raced a launch (arrival counters
// zeroed under a running kernel; decision-hook scratch zeroed while its gate kernels ran).  fill_now / copy_now enqueue on the
// given stream and wait for it, so they are ordered after everything queued there and complete on return;
// tests/test_abi_cpu.py fails on any other spelling in csrc/.
inline hipError_t fill_now(void* p, int v, size_t bytes, hipStream_t s) {
    const hipError_t e = hipMemsetAsync(p, v, bytes, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
inline hipError_t copy_now(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
inline hipError_t fill_async(void* p, int v, size_t bytes, hipStream_t s, const char* what) {
    if (g_trace) g_trace->lines.push_back("M " + g_trace->sname(s) + " " + what);
    return hipMemsetAsync(p, v, bytes, s);
}

template <typename F>
void launch(const Ctx& c, const char* kernel, double flops, double bytes, F&& f) {
    if (c.dry) return;
    if (g_trace) g_trace->lines.push_back("L " + g_trace->sname(c.s) + " " + kernel + (c.bwd_op ? std::string(" @") + c.bwd_op : std::string()));
    if (!c.prof) {
        HIPCHECK(f());
        return;
    }
    ProfRec r;
    r.kernel = kernel; r.op = c.prof->cur_op; r.flops = flops; r.bytes = bytes; r.phase = c.prof->phase;
    HIPCHECK(hipEventCreate(&r.e0));
    HIPCHECK(hipEventCreate(&r.e1));
    HIPCHECK(hipEventRecord(r.e0, c.s));
    HIPCHECK(f());
    HIPCHECK(hipEventRecord(r.e1, c.s));
    c.prof->recs.push_back(r);
}

std::atomic<int> g_live_handles{0};      // p3d_create .. p3d_destroy; p3d_shutdown refuses while any is alive

// ---- stream pool (round 5) ------------------------------------------------------------------------------------------------------
// A handle needs three streams (main and comm at the highest priority, side at the lowest).  A process that opens and closes many
// handles -- the test suite: ~250 sessions -- used to create and destroy three hardware queues per handle; the runtime releases
// them lazily, and one full-suite run of round 5 died with a bare abort() inside p3d_create after 235 tests (no message, not
// reproduced in two further runs).  Streams are now returned to a per-device, per-priority pool by p3d_destroy and taken from it
// by p3d_create: a long-lived process keeps three queues per device however many handles it has seen, and the K-slice scratch,
// which is keyed by stream, is reused with them instead of accumulating per session.  p3d_shutdown destroys the pooled streams.
struct StreamPool {
    std::mutex m;
    std::map<std::pair<int, int>, std::vector<hipStream_t>> idle;      // (device, priority class: 0 highest, 1 lowest)
};
StreamPool& stream_pool() { static StreamPool p; return p; }
hipStream_t take_stream(int device, int prio_class) {
    {
        std::lock_guard<std::mutex> g(stream_pool().m);
        auto& v = stream_pool().idle[{device, prio_class}];
        if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
    }
    int least = 0, greatest = 0;
    HIPCHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    HIPCHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_class == 0 ? greatest : least));
    return s;
}
void give_stream(int device, int prio_class, hipStream_t s) {      // the stream must be idle (p3d_destroy synchronises the device first)
    if (!s) return;
    std::lock_guard<std::mutex> g(stream_pool().m);
    stream_pool().idle[{device, prio_class}].push_back(s);
}
void destroy_pooled_streams() {
    std::lock_guard<std::mutex> g(stream_pool().m);
    for (auto& kv : stream_pool().idle)
        for (hipStream_t s : kv.second) hipStreamDestroy(s);
    stream_pool().idle.clear();
}
const float* g_zero_page = nullptr;      // 1 KiB of zeros (device), set by p3d_create / op entry points

void igemm_work(const IgemmArgs& a, double& flops, double& bytes) {
    const double M = (double)a.N * a.Gd * a.Gh * a.Gw;
    const double side = (double)a.N * a.Di * a.Hi * a.Wi;
    const double gathered = std::min(M * std::max(a.ntaps, 1), side);
    flops = 2.0 * M * a.ntaps * (double)a.K * a.Nc;
    bytes = 4.0 * (gathered * a.K + M * a.Nc * (1 + a.accum) + (double)a.ntaps * a.K * a.Nc);
}

void launch_igemm(const Ctx& c, const IgemmArgs& a0, int allow_split = 0) {
    IgemmArgs a = a0;
    double fl, by;
    igemm_work(a, fl, by);
    a.zeros = g_zero_page;
    const P3dIgemmPlan pl = p3d_igemm2_plan(a, allow_split);
    const char* name = pl.name;
    if (a.f16) name = pl.bm == 128 ? (pl.bn == 128 ? "igemm2_kernel<128,128,f16>" : "igemm2_kernel<128,64,f16>") : "igemm2_kernel<64,64,f16>";
    else if (a.at_mode || a.ngate) {      // fused-BatchNorm variants get their own rows in the per-kernel tables
        static const char* const tiles[3] = {"64,64", "128,64", "128,128"};
        static const char* const ats[4] = {"", ",relu1", ",relu2", ",bngrad"};
        static std::map<int, std::string> names;
        const int key = (pl.bm == 128 ? (pl.bn == 128 ? 2 : 1) : 0) * 8 + a.at_mode * 2 + (a.ngate ? 1 : 0);
        std::string& nm = names[key];
        if (nm.empty()) nm = std::string("igemm2_kernel<") + tiles[key / 8] + ats[a.at_mode] + (a.ngate ? ",gate" : "") + ">";
        name = nm.c_str();
    }
    launch(c, name, fl, by, [&]() { return p3d_launch_igemm2(a, pl, c.s); });
}

void zero_strided(const Ctx& c, float* p, int ld, int64_t rows, int C);

// Where a producer's BatchNorm-statistics epilogue puts its per-tile partial sums, and how many it wrote
// (read by p3d_bn_finalize right after, on the same stream).
struct StatSink { float* part = nullptr; int cap = 0; int* nparts = nullptr; };

// A group of implicit-GEMM launches that together produce one output tensor (one conv forward,
// or the residue classes of an input gradient / transposed conv).  Small problems slice K across blocks;
// the slices are folded in a fixed order by the last arriving block (conv_igemm2.hip), so the output needs no
// zero fill and the statistics epilogue and accumulate mode work either way.
// fork / join non-null: the launches are independent (residue classes of a transposed conv write disjoint output
// positions) and each of them leaves CUs idle (196 blocks of 128x128 on 256 CUs): odd ones go to the side stream so that
// two classes run at a time.
void run_igemm_group(const Ctx& c, std::vector<IgemmArgs>& v, float* out, int ld, int64_t rows, int C, bool accumulate,
                     const StatSink* stats, hipEvent_t fork = nullptr, hipEvent_t join = nullptr) {
    (void)out; (void)ld; (void)rows; (void)C;
    int base = 0;
    for (auto& a : v) {
        a.accum = accumulate ? 1 : 0;
        a.statpart = nullptr; a.stat_base = 0;
        a.zeros = g_zero_page;
        if (stats && stats->part) {
            const P3dIgemmPlan pl = p3d_igemm2_plan(a, 1);
            const int mt = p3d_igemm2_mtiles(a, pl);
            if (base + mt > stats->cap) throw P3dError("statistics partials overflow their arena slot");
            a.statpart = stats->part; a.stat_base = base;
            base += mt;
        }
    }
    // residue classes that share a plan go out as ONE launch (conv_igemm2.hip, igemm2_group_kernel): a class alone leaves
    // CUs idle (deconv3 at 8 clips: 196 tiles of 128x128 per class), all of them together fill the chip in a few waves
    if (v.size() >= 2 && !c.dry) {
        const P3dIgemmPlan pl = p3d_igemm2_plan(v[0], 1);
        if (p3d_igemm2_groupable(v.data(), (int)v.size(), pl)) {
            double fl = 0, by = 0;
            for (auto& a : v) { double f1, b1; igemm_work(a, f1, b1); fl += f1; by += b1; }
            const char* name = pl.bm == 128 ? (pl.bn == 128 ? "igemm2_group_kernel<128,128>" : "igemm2_group_kernel<128,64>") : "igemm2_group_kernel<64,64>";
            launch(c, name, fl, by, [&]() { return p3d_launch_igemm2_group(v.data(), (int)v.size(), pl, c.s); });
            if (stats && stats->nparts) *stats->nparts = base;
            return;
        }
    }
    const bool spread = fork && join && c.side && !c.dry && !c.prof && v.size() >= 4;
    Ctx sc = c;
    if (spread) {
        sc.s = c.side;
        HIPCHECK(ev_record(fork, c.s));
        HIPCHECK(ev_wait(c.side, fork));
    }
    int index = 0;
    for (auto& a : v) {
        launch_igemm((spread && (index & 1)) ? sc : c, a, 1);
        ++index;
    }
    if (spread) {
        HIPCHECK(ev_record(join, c.side));
        HIPCHECK(ev_wait(c.s, join));
    }
    if (stats && stats->nparts) *stats->nparts = base;
}

// Runs `f(side_ctx)` on the side stream after everything queued so far on the main stream.
template <typename F>
void on_side_stream(const Ctx& c, hipEvent_t ev, F&& f) {
    if (c.dry || !c.side || !ev) { f(c); return; }
    // timing diagnostic (WRONG RESULTS): P3D_TUNE_SKIP_SIDE=deconv,block1 drops the side-stream jobs queued during the backward of
    // ops whose name contains one of the substrings -- what that share of the filter gradients costs the step
    static const char* skip = [] {
        const char* e = p3d_tune_env("P3D_TUNE_SKIP_SIDE");
        if (e) fprintf(stderr, "[p3d] P3D_TUNE_SKIP_SIDE=%s: filter gradients are being DROPPED -- timing diagnostic, every result of this process is wrong\n", e);
        return e;
    }();
    if (skip && c.bwd_op) {
        std::string pats(skip), name(c.bwd_op);
        size_t a = 0;
        while (a <= pats.size()) {
            size_t b = pats.find(',', a);
            if (b == std::string::npos) b = pats.size();
            if (b > a && name.find(pats.substr(a, b - a)) != std::string::npos) return;
            a = b + 1;
        }
    }
    if (c.defer) { c.defer->emplace_back(ev, std::function<void(const Ctx&)>(f)); return; }   // f must own what it names
    HIPCHECK(ev_record(ev, c.s));
    HIPCHECK(ev_wait(c.side, ev));
    Ctx sc = c;
    sc.s = c.side;
    f(sc);
}

void launch_wgrad(const Ctx& c, const WgradArgs& a0) {
    WgradArgs a = a0;
    const double M = (double)a.N * a.Gd * a.Gh * a.Gw;
    const double side = (double)a.N * a.Di * a.Hi * a.Wi;
    const double fl = 2.0 * M * a.ntaps * (double)a.K * a.Nc;
    const double by = 4.0 * (std::min(M * a.ntaps, side) * a.K + M * a.Nc + (double)a.ntaps * a.K * a.Nc);
    a.zeros = g_zero_page;
    launch(c, p3d_wgrad2_variant(a), fl, by, [&]() { return p3d_launch_wgrad2(a, c.s); });
}

struct Op {
    std::string name, kind;
    double flops = 0, bytes = 0;            // forward algorithmic work
    double bflops = 0, bbytes = 0;          // backward algorithmic work
    std::vector<struct Param*> owns;        // trainable variables whose gradients this op's backward produces
    std::function<void(const Ctx&)> fwd, bwd;
    // Decisions of the last forward, for the decision-pinned parity tests (p3d_debug_decision_*): a normalise / ReLU pass runs
    // its OWN backward kernel on dz = 1 with the statistics terms off (the inference form dy = gamma*invstd * gate), so what
    // comes out is non-zero exactly where the gate of the real backward is open; a max-pool hands out its input.
    std::string dec_kind;                   // "" (none), "bn", "pool"
    std::string dec_name1, dec_name2;       // bn: TF scopes of the one or two BatchNorms
    const struct Act* dec_act = nullptr;    // bn: shape of the gated tensor; pool: the pool's input
    int64_t dec_scratch = 0;                // floats of scratch the gates closure needs (0: 4 x channels)
    std::function<void(hipStream_t, const float* ones, float* o1, float* o2, float* scratch)> gates;
};

struct ConvGeo {       // a SAME forward conv: input extents -> output extents
    int k[3], s[3], pad[3], I[3], O[3];
};

ConvGeo make_geo(int Di, int Hi, int Wi, const int k[3], const int s[3]) {
    ConvGeo g;
    const int in[3] = {Di, Hi, Wi};
    for (int a = 0; a < 3; ++a) {
        g.k[a] = k[a]; g.s[a] = s[a]; g.I[a] = in[a];
        g.O[a] = (in[a] + s[a] - 1) / s[a];
        int pt = (g.O[a] - 1) * s[a] + k[a] - in[a];
        if (pt < 0) pt = 0;
        g.pad[a] = pt / 2;
    }
    return g;
}

inline int pmod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// firstconv1 (p3d.py:172) on its packed form: the clip is copied to 4 channels with the SAME padding of the W axis written out
// (x4 [rows][Wp][4]), so that a kernel ROW is one tap of K = kw*4 contiguous floats (7 taps of K = 28 instead of 49 of K = 3).
struct StemGeo { int Wp, K4, KH; int64_t xrows; };
StemGeo stem_geo(const ConvGeo& g, int N) {
    if (g.k[0] != 1) throw P3dError("stem mode needs kd == 1");
    StemGeo sg;
    const int pad_total = std::max((g.O[2] - 1) * g.s[2] + g.k[2] - g.I[2], 0);
    sg.Wp = g.I[2] + pad_total; sg.K4 = g.k[2] * 4; sg.KH = g.k[1];
    sg.xrows = (int64_t)N * g.I[0] * g.I[1];
    return sg;
}
IgemmArgs stem_forward_args(const ConvGeo& g, int N, const StemGeo& sg, const float* x4, const float* w4, float* y, int ldy, int Cout,
                            const float* bias) {
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.Di = g.I[0]; a.Hi = g.I[1]; a.Wi = sg.Wp; a.ldx = 4; a.K = sg.K4;
    a.Gd = g.O[0]; a.Gh = g.O[1]; a.Gw = g.O[2]; a.isd = g.s[0]; a.ish = g.s[1]; a.isw = g.s[2];
    a.ntaps = sg.KH;
    for (int kh = 0; kh < sg.KH; ++kh) { a.taps[kh].dd = 0; a.taps[kh].dh = (int16_t)(kh - g.pad[1]); a.taps[kh].dw = 0; a.taps[kh].widx = (int16_t)kh; }
    a.x = x4; a.y = y; a.Do = g.O[0]; a.Ho = g.O[1]; a.Wo = g.O[2]; a.ldy = ldy; a.Nc = Cout;
    a.osd = a.osh = a.osw = 1; a.w = w4; a.bias = bias;
    return a;
}

// ---- launch-argument builders on the shared geometry ---------------------------------------------
IgemmArgs igemm_conv_forward(const ConvGeo& g, int N, const float* x, int ldx, int Cin, float* y, int ldy, int Cout,
                             const float* w, const float* bias, int accum, bool stem = false) {
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = N; a.Di = g.I[0]; a.Hi = g.I[1]; a.Wi = g.I[2]; a.ldx = ldx; a.K = Cin;
    a.Gd = g.O[0]; a.Gh = g.O[1]; a.Gw = g.O[2];
    a.isd = g.s[0]; a.ish = g.s[1]; a.isw = g.s[2];
    a.y = y; a.Do = g.O[0]; a.Ho = g.O[1]; a.Wo = g.O[2]; a.ldy = ldy; a.Nc = Cout;
    a.osd = a.osh = a.osw = 1;
    a.w = w; a.wT = 0; a.bias = bias; a.accum = accum;
    if (stem) throw P3dError("the 3-channel stem runs on its packed form (stem_forward_args)");
    int t = 0;
    for (int kd = 0; kd < g.k[0]; ++kd)
        for (int kh = 0; kh < g.k[1]; ++kh)
            for (int kw = 0; kw < g.k[2]; ++kw) {
                if (t >= P3D_MAX_TAPS) throw P3dError("kernel has too many taps");
                a.taps[t].dd = (int16_t)(kd - g.pad[0]);
                a.taps[t].dh = (int16_t)(kh - g.pad[1]);
                a.taps[t].dw = (int16_t)(kw - g.pad[2]);
                a.taps[t].widx = (int16_t)((kd * g.k[1] + kh) * g.k[2] + kw);
                ++t;
            }
    a.ntaps = t;
    return a;
}

// Input-gradient of the conv (== conv3d_transpose forward): one launch per residue class of the
// conv-input lattice.  `dense` has the conv's OUTPUT extents, `out` the conv's INPUT extents.
std::vector<IgemmArgs> igemm_conv_input_side(const ConvGeo& g, int N, const float* dense, int ld_dense, int Cdense,
                                             float* out, int ld_out, int Cout_side, const float* w,
                                             const float* bias, int accum, bool include_empty) {
    std::vector<IgemmArgs> v;
    for (int pd = 0; pd < g.s[0]; ++pd)
        for (int ph = 0; ph < g.s[1]; ++ph)
            for (int pw = 0; pw < g.s[2]; ++pw) {
                const int p[3] = {pd, ph, pw};
                IgemmArgs a;
                memset(&a, 0, sizeof(a));
                bool empty_grid = false;
                int G[3];
                for (int ax = 0; ax < 3; ++ax) {
                    G[ax] = (g.I[ax] - p[ax] + g.s[ax] - 1) / g.s[ax];
                    if (g.I[ax] <= p[ax]) empty_grid = true;
                }
                if (empty_grid) continue;
                a.x = dense; a.N = N; a.Di = g.O[0]; a.Hi = g.O[1]; a.Wi = g.O[2]; a.ldx = ld_dense; a.K = Cdense;
                a.Gd = G[0]; a.Gh = G[1]; a.Gw = G[2];
                a.isd = a.ish = a.isw = 1;
                a.y = out; a.Do = g.I[0]; a.Ho = g.I[1]; a.Wo = g.I[2]; a.ldy = ld_out; a.Nc = Cout_side;
                a.osd = g.s[0]; a.osh = g.s[1]; a.osw = g.s[2];
                a.ood = pd; a.ooh = ph; a.oow = pw;
                a.w = w; a.wT = 1; a.bias = bias; a.accum = accum;
                int t = 0;
                for (int kd = 0; kd < g.k[0]; ++kd) {
                    if (pmod(pd + g.pad[0] - kd, g.s[0])) continue;
                    for (int kh = 0; kh < g.k[1]; ++kh) {
                        if (pmod(ph + g.pad[1] - kh, g.s[1])) continue;
                        for (int kw = 0; kw < g.k[2]; ++kw) {
                            if (pmod(pw + g.pad[2] - kw, g.s[2])) continue;
                            if (t >= P3D_MAX_TAPS) throw P3dError("kernel has too many taps");
                            // exact division (the residue is 0): floor semantics for negatives
                            a.taps[t].dd = (int16_t)((pd + g.pad[0] - kd) / g.s[0]);
                            a.taps[t].dh = (int16_t)((ph + g.pad[1] - kh) / g.s[1]);
                            a.taps[t].dw = (int16_t)((pw + g.pad[2] - kw) / g.s[2]);
                            a.taps[t].widx = (int16_t)((kd * g.k[1] + kh) * g.k[2] + kw);
                            ++t;
                        }
                    }
                }
                a.ntaps = t;
                if (t == 0 && !include_empty) continue;
                v.push_back(a);
            }
    return v;
}

// Filter gradient of the [1,kh,kw,3,Cout] stem conv on its packed form (conv(): "stem"): x4 is the 4-channel, W-padded copy
// of the clip ([rows][Wp][4]), dw4 the packed gradient [kh][kw*4][Cout] (zeroed here); dw += its three real channels.
void stem_filter_gradient(const Ctx& c, const ConvGeo& g, int N, int Wp, const float* x4, const float* dy, int ldy, int Cout, float* dw4,
                          float* dw, float* dbias, bool greedy, float* onepass_part = nullptr, const BnBwdArgs* through_bn = nullptr) {
    const int KH = g.k[1], K4 = g.k[2] * 4;
    if (onepass_part && !dbias && p3d_stem_wgrad_ok(g.k[0], g.k[1], g.k[2], 3, Cout, g.s[0], g.s[1], g.s[2], g.O[2])) {
        // one pass over the output gradient (stem_wgrad.hip); through_bn: that gradient is the BatchNorm + ReLU OUTPUT's and
        // the normalisation's backward apply pass runs on this kernel's operand path
        StemWgradArgs a;
        memset(&a, 0, sizeof(a));
        a.x4 = x4; a.Wp = Wp; a.Hi = g.I[1]; a.nimg = N * g.I[0]; a.Ho = g.O[1]; a.Wo = g.O[2]; a.pad_h = g.pad[1];
        a.dy = dy; a.lddy = ldy; a.part = onepass_part; a.dw = dw;
        if (through_bn) {
            const BnBwdArgs& b = *through_bn;
            a.fused = 1; a.dy = b.dz; a.lddy = b.lddz; a.y = b.y1; a.ldy = b.ld1;
            a.scale = b.scale1; a.shift = b.shift1; a.mean = b.mean1; a.invstd = b.invstd1; a.gamma = b.gamma1; a.coef = b.coef1; a.batch = b.batch1;
        }
        const double rows = (double)a.nimg * a.Ho * a.Wo;
        int nblocks = 1;
        launch(c, through_bn ? "stem_wgrad_kernel<bn>" : "stem_wgrad_kernel", 2.0 * rows * KH * g.k[2] * 3 * Cout,
               4.0 * rows * Cout * (through_bn ? 2 : 1), [&]() { return p3d_stem_wgrad(a, &nblocks, c.s); });
        launch(c, "stem_wgrad_fold_kernel", 0, 4.0 * nblocks * KH * g.k[2] * 3 * Cout,
               [&]() { return p3d_stem_wgrad_fold(onepass_part, nblocks, dw, c.s); });
        return;
    }
    if (through_bn) throw P3dError("the stem's fused normalisation backward needs the one-pass filter gradient");
    if (!c.dry) HIPCHECK(hipMemsetAsync(dw4, 0, (size_t)KH * K4 * Cout * sizeof(float), c.s));
    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.x = x4; wa.N = N; wa.Di = g.I[0]; wa.Hi = g.I[1]; wa.Wi = Wp; wa.ldx = 4; wa.K = K4;
    wa.Gd = g.O[0]; wa.Gh = g.O[1]; wa.Gw = g.O[2]; wa.isd = g.s[0]; wa.ish = g.s[1]; wa.isw = g.s[2];
    wa.dy = dy; wa.ldy = ldy; wa.Nc = Cout; wa.dw = dw4; wa.dbias = dbias; wa.ksplit = 1;
    wa.ntaps = KH;
    for (int kh = 0; kh < KH; ++kh) { wa.taps[kh].dd = 0; wa.taps[kh].dh = (int16_t)(kh - g.pad[1]); wa.taps[kh].dw = 0; wa.taps[kh].widx = (int16_t)kh; }
    wa.greedy = greedy ? 1 : 0;
    wa.pair = K4 <= 32 ? 1 : 0;            // 28 floats per kernel row: two rows of the 7x7 kernel per 64-row tile
    launch_wgrad(c, wa);
    launch(c, "stem_unpack_dw_kernel", 0, 8.0 * KH * K4 * Cout, [&]() { return p3d_stem_unpack_dw(dw4, dw, KH * g.k[2], Cout, c.s); });
}

WgradArgs wgrad_conv(const ConvGeo& g, int N, const float* x, int ldx, int Cin, const float* dy, int ldy, int Cout,
                     float* dw, float* dbias, bool stem = false) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = N; a.Di = g.I[0]; a.Hi = g.I[1]; a.Wi = g.I[2]; a.ldx = ldx; a.K = Cin;
    a.Gd = g.O[0]; a.Gh = g.O[1]; a.Gw = g.O[2];
    a.isd = g.s[0]; a.ish = g.s[1]; a.isw = g.s[2];
    a.dy = dy; a.ldy = ldy; a.Nc = Cout; a.dw = dw; a.dbias = dbias; a.ksplit = 1;
    if (stem) throw P3dError("the 3-channel stem's filter gradient runs on its packed form (stem_filter_gradient)");
    int t = 0;
    for (int kd = 0; kd < g.k[0]; ++kd)
        for (int kh = 0; kh < g.k[1]; ++kh)
            for (int kw = 0; kw < g.k[2]; ++kw) {
                if (t >= P3D_MAX_TAPS) throw P3dError("kernel has too many taps");
                a.taps[t].dd = (int16_t)(kd - g.pad[0]);
                a.taps[t].dh = (int16_t)(kh - g.pad[1]);
                a.taps[t].dw = (int16_t)(kw - g.pad[2]);
                a.taps[t].widx = (int16_t)((kd * g.k[1] + kh) * g.k[2] + kw);
                ++t;
            }
    a.ntaps = t;
    return a;
}

// A plain row-major GEMM  Y[M x Nc] = X[M x K] * W  as a 1x1x1 convolution over one clip's lattice (D,H,W),
// M = D*H*W (the kernels pack lattice coordinates, so M is passed as the lattice it came from).
// W is [K][Nc] (wT = 0) or [Nc][K] (wT = 1), dense.
IgemmArgs gemm_rows(int D, int H, int W, const float* x, int ldx, int K, const float* w, int wT, float* y, int ldy, int Nc) {
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = 1; a.Di = D; a.Hi = H; a.Wi = W; a.ldx = ldx; a.K = K;
    a.Gd = D; a.Gh = H; a.Gw = W; a.isd = a.ish = a.isw = 1;
    a.y = y; a.Do = D; a.Ho = H; a.Wo = W; a.ldy = ldy; a.Nc = Nc; a.osd = a.osh = a.osw = 1;
    a.w = w; a.wT = wT;
    a.ntaps = 1;
    return a;
}
// dW[K x Nc] += X[M x K]^T * dY[M x Nc]  on the weight-gradient kernel (dW must be zero before)
WgradArgs gemm_tn(int D, int H, int W, const float* x, int ldx, int K, const float* dy, int ldy, int Nc, float* dw) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = 1; a.Di = D; a.Hi = H; a.Wi = W; a.ldx = ldx; a.K = K;
    a.Gd = D; a.Gh = H; a.Gw = W; a.isd = a.ish = a.isw = 1;
    a.dy = dy; a.ldy = ldy; a.Nc = Nc; a.dw = dw; a.ksplit = 1;
    a.ntaps = 1;
    return a;
}

void zero_strided(const Ctx& c, float* p, int ld, int64_t rows, int C) {
    if (c.dry) {
        if (ld == C) c.dry->push_back({p, (size_t)rows * C * sizeof(float)});      // dense buffers may move into the arena
        return;
    }
    if ((const char*)p >= c.z0 && (const char*)p < c.z1) return;                    // zeroed with the arena
    if (ld == C) HIPCHECK(hipMemsetAsync(p, 0, (size_t)rows * C * sizeof(float), c.s));
    else HIPCHECK(hipMemset2DAsync(p, (size_t)ld * sizeof(float), 0, (size_t)C * sizeof(float), (size_t)rows, c.s));
}

}  // namespace

namespace {
void ensure_zero_page() {
    // one page per process; igemm2 reads it for padded rows and channel tails
    if (g_zero_page) return;
    float* p = nullptr;
    HIPCHECK(hipMalloc((void**)&p, 1024));
    HIPCHECK(fill_now(p, 0, 1024, nullptr));      // process-wide, once, before any launch reads it
    g_zero_page = p;
}
}  // namespace

// ==================================================================================================
struct p3d_handle {
    p3d_config cfg;
    hipStream_t stream = nullptr, comm_stream = nullptr, side_stream = nullptr;
    bool side_pooled = true;                  // false: created with a CU mask (tuning builds), destroyed with the handle
    std::vector<hipEvent_t> fork_events;
    hipEvent_t ev_side_done = nullptr, ev_side_bucket = nullptr;
    // Events that only order this handle's own streams on one device: no system-scope fence when they complete (the default
    // writes the caches back for the host and for other devices -- tens of microseconds on the main stream at every hand-over
    // to the side stream; the events that gate the all-reduce and the host keep the default).
    static unsigned local_event_flags() {
        static const bool sysfence = [] { const char* e = p3d_tune_env("P3D_TUNE_EVENT_SYSFENCE"); return e && atoi(e); }();
        return hipEventDisableTiming | (sysfence ? 0u : (unsigned)hipEventDisableSystemFence);
    }
    hipEvent_t new_fork_event() {
        hipEvent_t e = nullptr;
        HIPCHECK(hipEventCreateWithFlags(&e, local_event_flags()));
        fork_events.push_back(e);
        return e;
    }
    std::vector<void*> allocs;

    std::deque<Param> params;                 // stable addresses
    std::map<std::string, Param*> pindex;
    std::vector<Param*> porder;               // creation order (trainables and states interleaved)
    int64_t n_train = 0, n_state = 0;         // floats in the flat buffers
    float *flat_p = nullptr, *flat_g = nullptr, *flat_m = nullptr, *flat_v = nullptr, *flat_state = nullptr;

    std::deque<Act> acts;
    std::map<std::string, Act*> named;
    std::deque<BN> bns;
    std::deque<GN> gns;
    std::deque<CbamSite> cbams;
    std::deque<char> flags;
    std::map<std::string, int> uniq;

    double* stats_arena = nullptr; int64_t stats_count = 0;
    double* red_arena = nullptr;   int64_t red_count = 0;
    float* bnbuf = nullptr;        int64_t bnbuf_count = 0;      // scale/shift/mean/invstd for every BN
    std::vector<std::function<void()>> late_bind;                // pointer fix-ups after arenas are allocated

    std::vector<Op> ops;
    char *zf = nullptr, *zb = nullptr;          // zero arenas: split-K outputs (forward) / gradients (backward)
    size_t zf_bytes = 0, zb_bytes = 0;
    Act* x_in = nullptr;
    Act* pred = nullptr;
    Act* logits = nullptr;
    float* d_y = nullptr;          // target
    float* d_dlogits = nullptr;
    double* d_loss = nullptr;
    float lr = 1e-4f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    int64_t step = 0;

    ncclComm_t comm = nullptr;
    hipEvent_t ev_bucket = nullptr, ev_comm_done = nullptr;
    int64_t bucket_floats = 8 << 20;          // 32 MB buckets

    // ---------------------------------------------------------------------------------------------
    template <typename T>
    T* dalloc(int64_t n) {
        void* p = nullptr;
        if (n <= 0) n = 1;
        HIPCHECK(hipMalloc(&p, (size_t)n * sizeof(T)));
        allocs.push_back(p);
        return (T*)p;
    }

    std::string unique(const std::string& base) {
        int k = uniq[base]++;
        return k == 0 ? base : base + "_" + std::to_string(k);
    }

    // BASELINE configs[4] option: 1x1x1 convolutions (forward and input gradient) round their operand fragments to
    // fp16 in registers and run on the fp16 MFMA with fp32 accumulation; storage, weights, statistics, every other
    // conv and all weight gradients stay fp32.  Off by default: the 1e-3 parity target is for the fp32 path.
    bool pointwise_f16 = false;
    std::string var_prefix;      // enclosing tf.variable_scope ("P3D/" for gn/p3d_gn.py:490), part of every variable name
    Param* add_param(const std::string& bare_name, std::vector<int64_t> shape, bool trainable, int init) {
        const std::string name = var_prefix + bare_name;
        if (pindex.count(name)) throw P3dError("duplicate variable " + name);
        params.emplace_back();
        Param* p = &params.back();
        p->name = name; p->shape = shape; p->trainable = trainable; p->init = init;
        p->count = 1;
        for (auto d : shape) p->count *= d;
        int64_t& total = trainable ? n_train : n_state;
        p->off = total;
        total += (p->count + 63) / 64 * 64;          // 256-byte aligned slots
        pindex[name] = p;
        porder.push_back(p);
        return p;
    }

    Act* new_act(const std::string& name, int N, int D, int H, int W, int C, bool with_grad = true) {
        acts.emplace_back();
        Act* a = &acts.back();
        a->name = name; a->N = N; a->D = D; a->H = H; a->W = W; a->C = C; a->ld = C;
        a->p = dalloc<float>(a->rows() * C);
        if (with_grad) a->g = dalloc<float>(a->rows() * C);
        if (!name.empty()) named[name] = a;
        return a;
    }
    Act* new_view(Act* parent, int coff, int C, const std::string& name) {
        acts.emplace_back();
        Act* a = &acts.back();
        *a = *parent;
        a->name = name; a->C = C; a->parent = parent; a->views.clear(); a->last_flag = nullptr;
        a->p = parent->p + coff;
        a->g = parent->g ? parent->g + coff : nullptr;
        parent->views.push_back(a);
        if (!name.empty()) named[name] = a;
        return a;
    }

    // Register a consumer of `a` whose backward adds into a->g.  Returns the flag the consumer reads
    // at backward time: 0 = first writer (overwrite), 1 = accumulate.  Backward runs consumers in
    // reverse registration order, so the newest registration is the writer.
    char* consume(Act* a) {
        flags.push_back(0);
        char* f = &flags.back();
        std::vector<Act*> region{a};
        // A concat buffer's whole-buffer consumer must be its newest one: its input gradient is then the
        // first write of every slice in backward order, and the slices' own consumers add to it.
        if (a->parent && a->parent->whole_consumed)
            throw P3dError("consumer of slice " + a->name + " registered after the consumer of its concat buffer " + a->parent->name);
        if (!a->views.empty()) a->whole_consumed = true;
        if (a->parent) region.push_back(a->parent);
        for (Act* v : a->views) region.push_back(v);
        for (Act* r : region) {
            if (r->last_flag) *r->last_flag = 1;
            r->last_flag = f;
        }
        return f;
    }

    // scratch of the per-sample BatchNorm inference path: two (sum, sumsq) tables and two scale/shift/mean/invstd
    // tables of [batch][widest BN]; ops run one after another on one stream, so they can share it
    double* ps_sums = nullptr; float* ps_tab = nullptr; int64_t ps_nc = 0;
    void ensure_per_sample_scratch() {
        if (ps_sums) return;
        int cmax = 4;
        for (auto& bn : bns) cmax = std::max(cmax, bn.C);
        ps_nc = (int64_t)cfg.batch * cmax;
        ps_sums = dalloc<double>(2 * 2 * ps_nc);
        ps_tab = dalloc<float>(2 * 4 * ps_nc);
    }
    BN* add_bn(const std::string& name_or_empty, int C, bool follows_flag) {
        bns.emplace_back();
        BN* bn = &bns.back();
        bn->name = name_or_empty.empty() ? unique("batch_normalization") : name_or_empty;
        bn->C = C; bn->follows_flag = follows_flag;
        bn->gamma = add_param(bn->name + "/gamma", {C}, true, INIT_ONES);
        bn->beta = add_param(bn->name + "/beta", {C}, true, INIT_ZEROS);
        bn->mm = add_param(bn->name + "/moving_mean", {C}, false, INIT_ZEROS);
        bn->mv = add_param(bn->name + "/moving_variance", {C}, false, INIT_ONES);
        const int64_t o = bnbuf_count; bnbuf_count += 4 * (int64_t)C;
        late_bind.push_back([this, bn, o, C]() {
            bn->scale = bnbuf + o; bn->shift = bnbuf + o + C; bn->mean = bnbuf + o + 2 * C; bn->invstd = bnbuf + o + 3 * C;
        });
        return bn;
    }
    // Reserve room for the producer's per-tile statistics partials of a `rows`-row output: one per 64-row tile of
    // every launch of the group (residue classes of a transposed conv: up to 64), or per block of p3d_bn_stats.
    float* statpart_arena = nullptr; int64_t statpart_count = 0;
    void reserve_stat_parts(BN* bn, int64_t rows) {
        if (bn->part_off >= 0) return;
        bn->part_cap = (int)std::max<int64_t>(rows / 64 + 80, p3d_bn_stats_parts((long)rows, bn->C));
        bn->part_off = statpart_count;
        statpart_count += (int64_t)bn->part_cap * bn->C * 2;
    }
    StatSink bn_sink(BN* bn) {
        if (bn->part_off < 0) throw P3dError("BatchNorm " + bn->name + " has no statistics arena slot");
        StatSink s; s.part = statpart_arena + bn->part_off; s.cap = bn->part_cap; s.nparts = &bn->nparts;
        return s;
    }
    // Producers of tensors that the one-launch small-tensor BN will consume need no statistics epilogue.
    static bool bn_is_small(int64_t rows, int C, bool dropout = false) { return !dropout && p3d_bn_small_ok((long)rows, C); }
    BN* stats_target(BN* bn, int64_t rows, int C, bool dropout = false) { return bn_is_small(rows, C, dropout) ? nullptr : bn; }
    BnParams bn_params(BN* bn) {
        BnParams b;
        b.gamma = bn->gamma->p; b.beta = bn->beta->p; b.moving_mean = bn->mm->p; b.moving_var = bn->mv->p;
        b.statpart = bn->part_off >= 0 ? statpart_arena + bn->part_off : nullptr; b.nparts = bn->nparts; b.scale = bn->scale; b.shift = bn->shift; b.mean = bn->mean; b.invstd = bn->invstd;
        b.C = bn->C;
        return b;
    }

    // ---- deferred, grouped weight gradients -------------------------------------------------------
    // A conv's filter gradient needs only its input and its output gradient, both of which stay untouched until the
    // step ends, and nothing waits for it before the all-reduce / optimiser.  So backward does not launch it on the
    // spot: problems queue up and go to the side stream several at a time (p3d_launch_wgrad2_group) -- the four
    // filter gradients of a stage-3 bottleneck offer 320 output tiles together, enough for the 256 CUs without
    // cutting the 784-position reduction.  Flushed when a group is full, before a gradient bucket is handed to the
    // all-reduce, and at the end of backward.
    struct PendingWgrad { WgradArgs a; std::string op; double flops, bytes; };
    std::vector<PendingWgrad> wq;
    std::vector<hipEvent_t> wq_events;          // one fork event per flush of a backward pass, reused every step
    int defer_release_op = -1;                  // backward: side-stream jobs of ops after this one wait until the walk reaches it
    double defer_budget = 0, parked_flops = 0;  // ... up to this many filter-gradient FLOPs (what the encoder's idle CUs can absorb)
    size_t wq_flushes = 0;
    static int64_t wgrad_tiles64(const WgradArgs& a) { return (int64_t)a.ntaps * ((a.K + 63) / 64) * ((a.Nc + 63) / 64); }
    void queue_wgrad(const Ctx& c, const WgradArgs& a0) {
        if (c.dry) return;
        WgradArgs a = a0;
        a.zeros = g_zero_page;
        const double M = (double)a.N * a.Gd * a.Gh * a.Gw;
        const double side = (double)a.N * a.Di * a.Hi * a.Wi;
        PendingWgrad pw;
        pw.a = a; pw.op = c.prof ? c.prof->cur_op : std::string();
        pw.flops = 2.0 * M * a.ntaps * (double)a.K * a.Nc;
        pw.bytes = 4.0 * (std::min(M * a.ntaps, side) * a.K + M * a.Nc + (double)a.ntaps * a.K * a.Nc);
        static const bool no_group = p3d_tune_env("P3D_NO_WGRAD_GROUP") != nullptr;
        static const int64_t flush_tiles = p3d_tune_env("P3D_WGRAD_FLUSH_TILES") ? atol(p3d_tune_env("P3D_WGRAD_FLUSH_TILES")) : 512;   // tuning: 256 -> 17.25 ms / step, 512 -> 17.0, 1024 with groups of 12 -> 17.1
        const bool alone = no_group || wgrad_tiles64(a) >= 256;      // fills the chip by itself (and may take 128x128 tiles)
        if (alone) flush_wgrads(c);
        wq.push_back(pw);
        int64_t tiles = 0;
        for (auto& q : wq) tiles += wgrad_tiles64(q.a);
        static const int group_max = p3d_tune_env("P3D_WGRAD_GROUP_MAX") ? std::max(1, std::min(P3D_WGRAD_GROUP, atoi(p3d_tune_env("P3D_WGRAD_GROUP_MAX")))) : P3D_WGRAD_GROUP;
        if (alone || (int)wq.size() >= group_max || tiles >= flush_tiles) flush_wgrads(c);
    }
    void flush_wgrads(const Ctx& c) {
        if (wq.empty() || c.dry) { wq.clear(); return; }
        if (wq_flushes >= wq_events.size()) {
            hipEvent_t e = nullptr;
            HIPCHECK(hipEventCreateWithFlags(&e, local_event_flags()));
            wq_events.push_back(e);
        }
        hipEvent_t ev = wq_events[wq_flushes++];
        std::vector<WgradArgs> probs;
        double fl = 0, by = 0;
        for (auto& q : wq) { probs.push_back(q.a); fl += q.flops; by += q.bytes; }
        bool any_fused = false;
        for (auto& pr : probs) any_fused |= pr.xt != 0 || pr.dyt != 0;
        const char* name = p3d_wgrad2_group_variant(probs.data(), (int)probs.size(), any_fused);
        if (c.defer) {            // parked: it will run beside the encoder's chain of small launches -- low residency (conv_wgrad2.hip)
            for (auto& pr : probs) pr.polite = 1;
            parked_flops += fl;
        }
        on_side_stream(c, ev, [=](const Ctx& sc) {          // by value: the job may be parked (Ctx::defer)
            launch(sc, name, fl, by, [&]() { return p3d_launch_wgrad2_group(probs.data(), (int)probs.size(), sc.s); });
        });
        wq.clear();
    }

    // ---- BatchNorm fused into the bottleneck convs' operand paths (p3d_kernels.h, conv_igemm2.hip) ---------------
    // The bn -> relu pairs INSIDE a bottleneck (p3d.py:56-81,88-97: after conv1, convS, convT) are not passes of their own
    // when fuse_bn is on: the consumer conv normalises its A fragments on the fly, the input-gradient launches gate and
    // reduce, the next input-gradient / filter-gradient launch applies BatchNorm's backward on its operand path.  The
    // unfused ops stay in the graph (per-sample inference statistics, p3d_set_bn_fusion(h, 0), parity tests of one
    // path against the other) and run instead when Ctx::fuse is off.
    // OFF by default: measured on MI355X at 8 clips of 16x112x112 (profiles/r03_bn_fusion_ab.json) the fused forward is a wash
    // (17.42-17.52 vs 17.33-17.36 ms / step, 96 launches fewer) and the fully fused step is slower (18.4-18.5 ms, 192 fewer):
    // with one wave per SIMD nothing hides the operand work, so it costs about what the removed launches did.
    bool fuse_bn = false;
    bool fuse_bn_bwd = false;         // p3d_set_bn_fusion(h, 2): the backward pass fused too
    bool last_forward_fused = false;
    // Only bottlenecks whose inner tensors have at most this many rows are built fusable: there a BatchNorm pass is a
    // latency-bound launch of its own (stage 3 at 8 clips of 16x112x112: 784 rows), while on big tensors the passes stream at
    // HBM speed and the convs are throughput-bound, so per-step operand work costs more than the passes it removes
    // (measured per stage).  P3D_FUSE_MAX_ROWS (read once, at p3d_create) overrides it for A/B runs.
    int64_t fuse_max_rows = 2048;
    static constexpr int FOLD_MAX = P3D_FOLD_MAX;      // up to this many partials per channel a consumer folds in its own prologue; beyond, a finalize launch
    struct FuseSrc { Act* y = nullptr; BN* bn = nullptr; int pub = 0; };   // pub: 0 read the published scale / shift, 1 fold the partials and publish, 2 fold only
    struct ConvFuse {
        int at = 0;                          // P3D_AT_RELU1 / P3D_AT_RELU2 on the conv's input
        FuseSrc src[2];
        int ngate = 0; FuseSrc gate[2];      // input-gradient epilogue: gated result -> gate.y->g, partial sums -> gate.bn
        Act* raw = nullptr; char* raw_flag = nullptr;     // ... and the raw result there (added to it when *raw_flag)
        bool accum_in = false;               // the raw gradient another consumer left in x->g is added before gating
        BN* out_bn = nullptr;                // the conv's output feeds a fused BatchNorm: dy = k1*g + k2*y + k3 on the operand paths
        bool any() const { return at != 0 || out_bn != nullptr; }
    };
    void make_fusable(BN* bn, int64_t rows) {
        if (bn->fusable) return;
        bn->fusable = true;
        reserve_stat_parts(bn, rows);
        bn->gpart_cap = (int)(rows / 64 + 8);
        bn->gpart_off = statpart_count;
        statpart_count += (int64_t)bn->gpart_cap * bn->C * 2;
        const int64_t o = bnbuf_count; bnbuf_count += 3 * (int64_t)bn->C;
        late_bind.push_back([this, bn, o]() { bn->coef = bnbuf + o; });
    }
    // forward: many partials -> one finalize launch that every consumer then reads (must precede a fork to the side stream)
    std::map<BN*, bool> fwd_finalized, grad_finalized;      // per pass: the published values are complete
    void fused_prefinalize(const Ctx& c, const ConvFuse& cf, int64_t rows) {
        if (c.dry) return;
        for (int q = 0; q < (cf.at == P3D_AT_RELU2 ? 2 : 1); ++q) {
            BN* bn = cf.src[q].bn;
            if (cf.src[q].pub != 1) continue;
            fwd_finalized[bn] = false;
            bn->used_batch = true;
            if (bn->nparts > FOLD_MAX) {
                launch(c, "bn_finalize_kernel", 0, 64.0 * bn->C, [&]() { return p3d_bn_finalize(bn_params(bn), (long)rows, 1, c.update_moving ? 1 : 0, 1e-3f, c.s); });
                fwd_finalized[bn] = true;
            }
        }
    }
    BnFold bn_fold(const FuseSrc& s, int64_t rows, const Ctx& c) {
        BN* bn = s.bn;
        BnFold f;
        memset(&f, 0, sizeof(f));
        f.gamma = bn->gamma->p; f.beta = bn->beta->p; f.C = bn->C;
        f.scale = bn->scale; f.shift = bn->shift; f.mean = bn->mean; f.invstd = bn->invstd;
        f.moving_mean = bn->mm->p; f.moving_var = bn->mv->p;
        f.inv_m = 1.0 / (double)rows; f.eps = 1e-3f;
        const bool fold = s.pub != 0 && !fwd_finalized[bn];
        if (fold) { f.part = statpart_arena + bn->part_off; f.nparts = bn->nparts; }
        f.publish = (fold && s.pub == 1) ? 1 : 0;
        f.update_moving = c.update_moving ? 1 : 0;
        return f;
    }
    BnGradFold bn_grad_fold(BN* bn, int64_t rows, bool publish) {
        BnGradFold f;
        memset(&f, 0, sizeof(f));
        f.gamma = bn->gamma->p; f.mean = bn->mean; f.invstd = bn->invstd; f.C = bn->C;
        f.coef = bn->coef; f.dgamma = bn->gamma->g; f.dbeta = bn->beta->g;
        f.inv_m = 1.0 / (double)rows;
        const bool fold = !grad_finalized[bn];
        if (fold) { f.part = statpart_arena + bn->gpart_off; f.nparts = bn->gnparts; }
        f.publish = (fold && publish) ? 1 : 0;
        return f;
    }
    BnGate bn_gate(const FuseSrc& s) {
        BnGate g;
        memset(&g, 0, sizeof(g));
        g.y = s.y->p; g.ldy = s.y->ld;
        g.scale = s.bn->scale; g.shift = s.bn->shift; g.mean = s.bn->mean; g.invstd = s.bn->invstd;
        g.out = s.y->g; g.ldo = s.y->ld;
        g.part = statpart_arena + s.bn->gpart_off;
        return g;
    }

    std::vector<IgemmArgs> sib_pending;      // a sibling pair's first launch, waiting for the second (conv(): sibling)

#include "net_ops.inc"
#include "net_gn.inc"
#include "net_graphs.inc"
#include "net_plan.inc"
#include "net_sched.inc"
};

#include "net_abi.inc"
