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

template <typename F>
void launch(const Ctx& c, const char* kernel, double flops, double bytes, F&& f) {
    if (c.dry) return;
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
        HIPCHECK(hipEventRecord(fork, c.s));
        HIPCHECK(hipStreamWaitEvent(c.side, fork, 0));
    }
    int index = 0;
    for (auto& a : v) {
        launch_igemm((spread && (index & 1)) ? sc : c, a, 1);
        ++index;
    }
    if (spread) {
        HIPCHECK(hipEventRecord(join, c.side));
        HIPCHECK(hipStreamWaitEvent(c.s, join, 0));
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
    HIPCHECK(hipEventRecord(ev, c.s));
    HIPCHECK(hipStreamWaitEvent(c.side, ev, 0));
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
                          float* dw, float* dbias, bool greedy) {
    const int KH = g.k[1], K4 = g.k[2] * 4;
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
    HIPCHECK(hipMemset(p, 0, 1024));
    g_zero_page = p;
}
}  // namespace

// ==================================================================================================
struct p3d_handle {
    p3d_config cfg;
    hipStream_t stream = nullptr, comm_stream = nullptr, side_stream = nullptr;
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
        const char* name = probs.size() == 1 ? p3d_wgrad2_variant(probs[0]) : (any_fused ? "wgrad2_kernel<64,64,fused>(grouped)" : "wgrad2_kernel<64,64>(grouped)");
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

    // ---- graph ops ---------------------------------------------------------------------------
    // tf.nn.conv3d / tf.layers.conv3d: SAME conv, optional bias, optional BN-statistics epilogue.
    Act* conv(const std::string& opname, Act* x, Param* w, Param* bias, const int k[3], const int s[3], int Cout, BN* bn,
              const std::string& out_name, bool stem = false, bool bn_has_dropout = false, int sibling = 0,
              const ConvFuse* fuse = nullptr) {
        const ConvGeo g = make_geo(x->D, x->H, x->W, k, s);
        Act* y = new_act(out_name, x->N, g.O[0], g.O[1], g.O[2], Cout);
        if (bn && stats_target(bn, y->rows(), Cout, bn_has_dropout)) reserve_stat_parts(bn, y->rows());
        const ConvFuse cf = fuse ? *fuse : ConvFuse();
        if (cf.out_bn) { if (cf.out_bn != bn) throw P3dError("fused output BatchNorm must be the conv's own"); make_fusable(bn, y->rows()); }
        if (cf.ngate && (s[0] != 1 || s[1] != 1 || s[2] != 1)) throw P3dError("gated input gradients need a stride-1 conv");
        char* xflag = x->g ? consume(x) : nullptr;
        const int Cin = x->C;
        const int ntap = k[0] * k[1] * k[2];
        Op op;
        op.name = opname; op.kind = stem ? "conv_stem" : (ntap == 1 ? "conv_1x1x1" : "conv_kxkxk");
        op.flops = 2.0 * y->rows() * ntap * Cin * Cout;
        op.bytes = 4.0 * (x->rows() * (double)Cin + y->rows() * (double)Cout + (double)ntap * Cin * Cout);
        op.bflops = op.flops * (x->g ? 2 : 1);
        op.bbytes = op.bytes * (x->g ? 2 : 1);
        op.owns = {w}; if (bias) op.owns.push_back(bias);
        // a fused BatchNorm's parameter gradients come out of this conv's input-gradient launch (bn_grad_fold_channel)
        if (cf.out_bn) { op.owns.push_back(cf.out_bn->gamma); op.owns.push_back(cf.out_bn->beta); }
        hipEvent_t fork_ev = new_fork_event();
        if (stem) {
            // firstconv1 (p3d.py:172) on the pipelined kernels: 4-channel, W-padded copy of the clip, kw*4 contiguous
            // floats per kernel row (elementwise.hip, "stem"); 7 taps of K = 28 instead of 49 taps of K = 3
            if (k[0] != 1 || Cin != 3 || bias) throw P3dError("stem path is for [1,kh,kw,3,C] kernels without bias");
            const StemGeo sg = stem_geo(g, x->N);
            const int Wp = sg.Wp, K4 = sg.K4, KH = sg.KH;
            const int64_t xrows = sg.xrows;
            float* x4 = dalloc<float>(xrows * Wp * 4);
            HIPCHECK(hipMemset(x4, 0, (size_t)xrows * Wp * 4 * sizeof(float)));
            float* w4 = dalloc<float>((int64_t)KH * K4 * Cout);
            float* dw4 = dalloc<float>((int64_t)KH * K4 * Cout);
            op.fwd = [=](const Ctx& c) {
                launch(c, "stem_pad_kernel", 0, 28.0 * x->rows(), [&]() { return p3d_stem_pad(x->p, x4, xrows, g.I[2], Wp, g.pad[2], c.s); });
                launch(c, "stem_pack_w_kernel", 0, 8.0 * KH * K4 * Cout, [&]() { return p3d_stem_pack_w(w->p, w4, KH * g.k[2], Cout, c.s); });
                const IgemmArgs a = stem_forward_args(g, x->N, sg, x4, w4, y->p, y->ld, Cout, nullptr);
                std::vector<IgemmArgs> v{a};
                BN* sbn = bn ? stats_target(bn, y->rows(), Cout, bn_has_dropout) : nullptr;
                StatSink sink; if (sbn) sink = bn_sink(sbn);
                run_igemm_group(c, v, y->p, y->ld, y->rows(), Cout, false, sbn ? &sink : nullptr);
            };
            op.bwd = [=](const Ctx& c) {
                on_side_stream(c, fork_ev, [=](const Ctx& sc) {
                    // greedy: the last launch of the backward pass -- the main stream is done
                    stem_filter_gradient(sc, g, x->N, Wp, x4, y->g, y->ld, Cout, dw4, w->g, nullptr, /*greedy=*/true);
                });
                if (xflag) throw P3dError("the stem input carries no gradient");
            };
            ops.push_back(op);
            return y;
        }
        // sibling convs on one input (ST_B: convS and convT both read relu(bn1(.)), p3d.py:65-72): the first one only prepares
        // its launch, the second sends both out as ONE grouped launch (conv_igemm2.hip, igemm2_group_kernel) -- either alone
        // leaves most CUs idle, and forking one to the side stream costs ~10 us of cross-stream latency each way
        auto sibling_prepare = [=](const Ctx& c, IgemmArgs& a) {
            a.zeros = g_zero_page; a.accum = 0; a.statpart = nullptr; a.stat_base = 0;
            BN* sbn = bn ? ((c.fuse && cf.out_bn) ? bn : stats_target(bn, y->rows(), Cout, bn_has_dropout)) : nullptr;
            if (sbn) {
                const StatSink sink = bn_sink(sbn);
                const int mt = p3d_igemm2_mtiles(a, p3d_igemm2_plan(a, 1));
                if (mt > sink.cap) throw P3dError("statistics partials overflow their arena slot");
                a.statpart = sink.part; *sink.nparts = mt;
            }
        };
        auto fwd_body = [=](const Ctx& c) {
            const bool fz = c.fuse && cf.at != 0;
            // fused: the A operand is the raw output of the conv before the BatchNorm (src[0].y), normalised on the fly
            const Act* xs = fz ? cf.src[0].y : x;
            std::vector<IgemmArgs> v{igemm_conv_forward(g, x->N, xs->p, xs->ld, Cin, y->p, y->ld, Cout, w->p, bias ? bias->p : nullptr,
                                                        0, stem)};
            if (fz) {
                IgemmArgs& a = v[0];
                a.at_mode = cf.at;
                a.f1 = bn_fold(cf.src[0], cf.src[0].y->rows(), c);
                if (cf.at == P3D_AT_RELU2) { a.x2 = cf.src[1].y->p; a.ldx2 = cf.src[1].y->ld; a.f2 = bn_fold(cf.src[1], cf.src[1].y->rows(), c); }
            }
            if (ntap == 1 && !stem && pointwise_f16) v[0].f16 = 1;
            // a fused BatchNorm behind this conv always needs the tile partials (the one-launch small-tensor BN does not)
            BN* sbn = bn ? ((c.fuse && cf.out_bn) ? bn : stats_target(bn, y->rows(), Cout, bn_has_dropout)) : nullptr;
            StatSink sink; if (sbn) sink = bn_sink(sbn);
            run_igemm_group(c, v, y->p, y->ld, y->rows(), Cout, false, sbn ? &sink : nullptr);
        };
        // sibling = 1 / 2: first / second of two convs that read the same input (ST_B, p3d.py:65-72)
        hipEvent_t fork_fwd = nullptr;
        op.fwd = [=](const Ctx& c) {
            if (c.fuse && cf.at) fused_prefinalize(c, cf, cf.src[0].y->rows());      // on the main stream, ahead of a fork
            (void)fork_fwd;
            if (sibling && !c.dry && !(c.fuse && cf.at) && ntap > 0 && !stem) {
                IgemmArgs a = igemm_conv_forward(g, x->N, x->p, x->ld, Cin, y->p, y->ld, Cout, w->p, bias ? bias->p : nullptr, 0, false);
                sibling_prepare(c, a);
                if (sibling == 1) {       // first of the pair: wait for the second (a stale entry would be a third class of the next pair)
                    if (!sib_pending.empty()) throw P3dError("sibling conv " + opname + ": an earlier pair never sent its launch");
                    sib_pending.push_back(a);
                    return;
                }
                if (sib_pending.size() != 1) throw P3dError("sibling conv " + opname + " has no first sibling waiting");
                sib_pending.push_back(a);
                std::vector<IgemmArgs> v;
                v.swap(sib_pending);
                const P3dIgemmPlan pl = p3d_igemm2_plan(v[0], 1);
                if (p3d_igemm2_groupable(v.data(), (int)v.size(), pl)) {
                    double fl = 0, by = 0;
                    for (auto& q : v) { double f1, b1; igemm_work(q, f1, b1); fl += f1; by += b1; }
                    const std::string kn = std::string("igemm2_group_kernel<") + std::to_string(pl.bm) + "," + std::to_string(pl.bn) + ">(siblings)";
                    launch(c, kn.c_str(), fl, by, [&]() { return p3d_launch_igemm2_group(v.data(), (int)v.size(), pl, c.s); });
                } else {
                    for (auto& q : v) launch_igemm(c, q, 1);
                }
                return;
            }
            fwd_body(c);
        };
        op.bwd = [=](const Ctx& c) {
            if (!(c.fuse_bwd && cf.any())) {
                // BatchNorm's backward as launches of its own.  After a FUSED forward the normalised input was never stored:
                // the filter gradient reads it as relu(scale*y + shift) on its operand path (scale / shift published by the forward)
                const bool fin = c.fuse && cf.at != 0;
                WgradArgs wa = wgrad_conv(g, x->N, (fin ? cf.src[0].y : x)->p, (fin ? cf.src[0].y : x)->ld, Cin, y->g, y->ld, Cout, w->g,
                                          bias ? bias->g : nullptr, stem);
                if (fin) {
                    wa.xt = cf.at == P3D_AT_RELU2 ? 2 : 1;
                    wa.xs1 = cf.src[0].bn->scale; wa.xt1 = cf.src[0].bn->shift;
                    if (wa.xt == 2) { wa.x2 = cf.src[1].y->p; wa.ldx2 = cf.src[1].y->ld; wa.xs2 = cf.src[1].bn->scale; wa.xt2 = cf.src[1].bn->shift; }
                }
                queue_wgrad(c, wa);
                if (xflag) {
                    const int accum = *xflag;
                    auto v = igemm_conv_input_side(g, x->N, y->g, y->ld, Cout, x->g, x->ld, Cin, w->p, nullptr, accum,
                                                   /*include_empty=*/!accum);
                    if (ntap == 1 && !stem && pointwise_f16) for (auto& a : v) a.f16 = 1;
                    run_igemm_group(c, v, x->g, x->ld, x->rows(), Cin, accum != 0, nullptr);
                }
                return;
            }
            // ---- fused BatchNorm: y->g holds the GATED gradient of relu(bn(y)) (written by the consumer's input-gradient
            //      launch), BatchNorm's own backward happens on the operand paths below
            const bool fin = cf.at != 0;
            WgradArgs wa = wgrad_conv(g, x->N, (fin ? cf.src[0].y : x)->p, (fin ? cf.src[0].y : x)->ld, Cin, y->g, y->ld, Cout, w->g,
                                      bias ? bias->g : nullptr, stem);
            if (fin) {
                wa.xt = cf.at == P3D_AT_RELU2 ? 2 : 1;
                wa.xs1 = cf.src[0].bn->scale; wa.xt1 = cf.src[0].bn->shift;
                if (wa.xt == 2) { wa.x2 = cf.src[1].y->p; wa.ldx2 = cf.src[1].y->ld; wa.xs2 = cf.src[1].bn->scale; wa.xt2 = cf.src[1].bn->shift; }
            }
            if (cf.out_bn) { wa.dyt = 1; wa.dy2 = y->p; wa.ldy2 = y->ld; wa.dcoef = cf.out_bn->coef; }
            if (!xflag) throw P3dError("a conv with a fused BatchNorm needs an input gradient launch (it publishes the coefficients)");
            if (cf.out_bn && !c.dry) {
                grad_finalized[cf.out_bn] = false;
                if (cf.out_bn->gnparts > FOLD_MAX) {
                    const BnGradFold gf = bn_grad_fold(cf.out_bn, y->rows(), true);
                    launch(c, "bn_grad_finalize_kernel", 0, 64.0 * cf.out_bn->C, [&]() { return p3d_bn_grad_finalize(gf, c.s); });
                    grad_finalized[cf.out_bn] = true;
                }
            }
            // where the input gradient goes: plain inputs and ungated launches write / add to x->g; gated launches send the
            // gated result to the gate's own buffers and use x->g (or cf.raw) for raw partial results only
            float* py = x->g; int pld = x->ld; int accum = *xflag;
            if (cf.ngate) {
                accum = cf.accum_in ? (int)*xflag : 0;
                if (cf.raw) { py = cf.raw->g; pld = cf.raw->ld; accum = *cf.raw_flag; }
            }
            auto v = igemm_conv_input_side(g, x->N, y->g, y->ld, Cout, py, pld, Cin, w->p, nullptr, accum, /*include_empty=*/!accum);
            bool first = true;
            for (auto& a : v) {
                if (ntap == 1 && !stem && pointwise_f16) a.f16 = 1;
                if (cf.out_bn) {
                    a.at_mode = P3D_AT_GRAD; a.x2 = y->p; a.ldx2 = y->ld;
                    a.gf = bn_grad_fold(cf.out_bn, y->rows(), first);
                }
                if (cf.ngate) {
                    if (v.size() != 1) throw P3dError("gated input gradient with more than one residue class");
                    a.ngate = cf.ngate; a.raw_store = cf.raw ? 1 : 0;
                    for (int q = 0; q < cf.ngate; ++q) a.gate[q] = bn_gate(cf.gate[q]);
                    if (!c.dry) {
                        IgemmArgs t = a; t.zeros = g_zero_page;
                        const int mt = p3d_igemm2_mtiles(t, p3d_igemm2_plan(t, 1));
                        for (int q = 0; q < cf.ngate; ++q) {
                            if (mt > cf.gate[q].bn->gpart_cap) throw P3dError("gradient partials overflow their arena slot");
                            cf.gate[q].bn->gnparts = mt;
                        }
                    }
                }
                first = false;
            }
            run_igemm_group(c, v, py, pld, x->rows(), Cin, accum != 0, nullptr);
            queue_wgrad(c, wa);      // after the input gradient: its block 0 published the coefficients this one reads
        };
        ops.push_back(op);
        return y;
    }

    // tf.layers.conv3d_transpose(x, filters, k, s, 'same'): kernel [kd,kh,kw,Cout,Cin].
    Act* deconv(const std::string& opname, Act* x, Param* kern, Param* bias, const int k[3], const int s[3], int Cout, BN* bn,
                const std::string& out_name, bool bn_has_dropout = false) {
        const ConvGeo g = make_geo(x->D * s[0], x->H * s[1], x->W * s[2], k, s);    // conv whose input is y
        Act* y = new_act(out_name, x->N, g.I[0], g.I[1], g.I[2], Cout);
        if (bn && stats_target(bn, y->rows(), Cout, bn_has_dropout)) reserve_stat_parts(bn, y->rows());
        char* xflag = x->g ? consume(x) : nullptr;
        const int Cin = x->C;
        Op op;
        op.name = opname; op.kind = "deconv";
        double taps_eff = 1;
        for (int a = 0; a < 3; ++a) taps_eff *= (double)k[a] / s[a];
        op.flops = 2.0 * y->rows() * taps_eff * Cin * Cout;
        op.bytes = 4.0 * (x->rows() * (double)Cin + y->rows() * (double)Cout + (double)k[0] * k[1] * k[2] * Cin * Cout);
        op.bflops = 2 * op.flops; op.bbytes = 2 * op.bytes;
        op.owns = {kern}; if (bias) op.owns.push_back(bias);
        hipEvent_t fork_ev = new_fork_event();
        hipEvent_t class_fork = new_fork_event(), class_join = new_fork_event();
        op.fwd = [=](const Ctx& c) {
            auto v = igemm_conv_input_side(g, x->N, x->p, x->ld, Cin, y->p, y->ld, Cout, kern->p, bias ? bias->p : nullptr,
                                           0, true);
            BN* sbn = bn ? stats_target(bn, y->rows(), Cout, bn_has_dropout) : nullptr;
            StatSink sink; if (sbn) sink = bn_sink(sbn);
            run_igemm_group(c, v, y->p, y->ld, y->rows(), Cout, false, sbn ? &sink : nullptr, class_fork, class_join);
        };
        op.bwd = [=](const Ctx& c) {
            // dK[tap][co][ci] = sum dy_big[o][co] * x[i][ci]  (conv wgrad with the roles of x and dy swapped)
            queue_wgrad(c, wgrad_conv(g, x->N, y->g, y->ld, Cout, x->p, x->ld, Cin, kern->g, nullptr));
            if (bias)
                on_side_stream(c, fork_ev, [=](const Ctx& sc) {
                    launch(sc, "colsum_kernel", 0, 4.0 * y->rows() * Cout, [&]() { return p3d_colsum(y->g, y->ld, y->rows(), Cout, bias->g, sc.s); });
                });
            if (xflag) {
                std::vector<IgemmArgs> v{igemm_conv_forward(g, x->N, y->g, y->ld, Cout, x->g, x->ld, Cin, kern->p, nullptr, *xflag)};
                run_igemm_group(c, v, x->g, x->ld, x->rows(), Cin, *xflag != 0, nullptr);
            }
        };
        ops.push_back(op);
        return y;
    }

    // BN finalize + fused normalise / ReLU / add pass (modes in p3d_kernels.h) and its backward.
    // fused_site: with BatchNorm fusion on (Ctx::fuse) this pass does not run -- its consumers normalise on their operand
    // paths and the producing convs own the parameter gradients (conv(): ConvFuse); it runs when fusion is off.
    Act* bn_apply(const std::string& opname, int mode, Act* y1, BN* bn1, Act* y2, BN* bn2, Act* out, const std::string& out_name,
                  bool dropout = false, bool fused_site = false) {
        if (!out) out = new_act(out_name, y1->N, y1->D, y1->H, y1->W, y1->C);
        else if (!out_name.empty()) named[out_name] = out;
        consume(y1);                                 // y1 is a raw conv output: this op is its only consumer
        char* f2 = nullptr;
        if (y2) f2 = consume(y2);
        const bool two = (mode == 2 || mode == 3);
        const int bwd_parts = p3d_bn_bwd_parts((long)y1->rows(), y1->C);
        const int64_t red_off = statpart_count;          // backward partial sums share the (never zeroed) partials arena
        statpart_count += (two ? 2 : 1) * (int64_t)bwd_parts * 2 * y1->C;
        const int64_t coef_off = bnbuf_count;
        bnbuf_count += (two ? 4 : 2) * (int64_t)y1->C;
        const int64_t M = y1->rows();
        const int C = y1->C;
        Op op;
        op.name = opname; op.kind = "bn_apply" + std::to_string(mode);
        const double tens = (double)M * C * 4.0;
        const std::string kn_apply = "bn_apply_kernel<" + std::to_string(mode) + ">";
        const std::string kn_red = "bn_bwd_reduce_kernel<" + std::to_string(mode) + ">";
        const std::string kn_bapply = "bn_bwd_apply_kernel<" + std::to_string(mode) + ">";
        op.flops = 0; op.bytes = tens * (y2 ? 3 : 2);
        op.bflops = 0; op.bbytes = tens * (y2 ? 7 : 5);
        if (!fused_site) {
            op.owns = {bn1->gamma, bn1->beta};
            if (two) { op.owns.push_back(bn2->gamma); op.owns.push_back(bn2->beta); }
        } else {
            if (mode != 0 && mode != 3 && mode != 4) throw P3dError("only the bn -> relu passes inside a bottleneck fuse");
            // reading the (never stored) normalised tensor after a fused forward: the plain apply pass on the published tables
            out->materialize = [=](hipStream_t st) {
                BnApplyArgs a;
                memset(&a, 0, sizeof(a));
                a.mode = mode; a.M = M; a.C = C;
                a.y1 = y1->p; a.ld1 = y1->ld; a.scale1 = bn1->scale; a.shift1 = bn1->shift;
                if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; }
                if (two) { a.scale2 = bn2->scale; a.shift2 = bn2->shift; }
                a.z = out->p; a.ldz = out->ld;
                if (mode == 4 && y2->materialize) y2->materialize(st);      // ST_C adds relu(bnS(yS)), itself never stored
                HIPCHECK(p3d_bn_apply(a, st));
            };
        }
        const bool small = bn_is_small(M, C, dropout);
        auto small_args = [=](const Ctx& c) {
            BnSmallArgs a;
            memset(&a, 0, sizeof(a));
            a.mode = mode; a.M = (int)M; a.C = C;
            a.y1 = y1->p; a.ld1 = y1->ld;
            if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; }
            a.bn1 = bn_params(bn1);
            if (two) a.bn2 = bn_params(bn2);
            a.update_moving = c.update_moving; a.eps = 1e-3f;
            a.z = out->p; a.ldz = out->ld;
            a.dz = out->g; a.lddz = out->ld;
            a.dy1 = y1->g; a.lddy1 = y1->ld;
            if (y2) { a.dy2 = y2->g; a.lddy2 = y2->ld; a.acc2 = *f2; }
            a.dgamma1 = bn1->gamma->g; a.dbeta1 = bn1->beta->g;
            if (two) { a.dgamma2 = bn2->gamma->g; a.dbeta2 = bn2->beta->g; }
            return a;
        };
        const std::string kn_sf = "bn_small_fwd_kernel<" + std::to_string(mode) + ">";
        const std::string kn_sb = "bn_small_bwd_kernel<" + std::to_string(mode) + ">";
        const int R = y1->D * y1->H * y1->W;
        op.fwd = [=](const Ctx& c) {
            if (fused_site && c.fuse) return;
            if (c.per_sample && (bn1->follows_flag ? c.training : true)) {
                // B independent batch-of-1 normalisations: per-(clip, channel) statistics over D*H*W, i.e. the
                // GroupNorm machinery with one channel per group and BN's epsilon
                if (c.training || c.update_moving || dropout) throw P3dError("per-sample BatchNorm is an inference path");
                ensure_per_sample_scratch();
                auto norm = [&](BN* bn, Act* y, int slot) {
                    GnParams p;
                    memset(&p, 0, sizeof(p));
                    const int64_t nc = (int64_t)y->N * C;
                    p.gamma = bn->gamma->p; p.beta = bn->beta->p; p.C = C; p.G = C;
                    p.sums = ps_sums + (int64_t)slot * 2 * ps_nc;
                    float* t = ps_tab + (int64_t)slot * 4 * ps_nc;
                    p.scale = t; p.shift = t + nc; p.mean = t + 2 * nc; p.invstd = t + 3 * nc;
                    if (!c.dry) HIPCHECK(hipMemsetAsync(p.sums, 0, (size_t)nc * 2 * sizeof(double), c.s));
                    launch(c, "gn_stats_kernel", 0, tens, [&]() { return p3d_gn_stats(y->p, y->ld, y->N, R, C, p.sums, c.s); });
                    launch(c, "gn_finalize_kernel", 0, 32.0 * nc, [&]() { return p3d_gn_finalize(p, y->N, R, 1e-3f, c.s); });
                    return p;
                };
                GnApplyArgs a;
                memset(&a, 0, sizeof(a));
                a.mode = mode; a.M = M; a.R = R; a.C = C;
                a.y1 = y1->p; a.ld1 = y1->ld; a.g1 = norm(bn1, y1, 0);
                if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; }
                if (two) a.g2 = norm(bn2, y2, 1);
                a.z = out->p; a.ldz = out->ld;
                launch(c, "gn_apply_kernel(per-sample BN)", 0, tens * (y2 ? 3 : 2), [&]() { return p3d_gn_apply(a, c.s); });
                return;
            }
            if (small) {
                bn1->used_batch = bn1->follows_flag ? c.training : true;
                if (two) bn2->used_batch = bn2->follows_flag ? c.training : true;
                BnSmallArgs a = small_args(c);
                a.batch1 = bn1->used_batch; a.batch2 = two ? bn2->used_batch : 0;
                launch(c, kn_sf.c_str(), 0, tens * (y2 ? 3 : 2), [&]() { return p3d_bn_small_fwd(a, c.s); });
                return;
            }
            auto fin = [&](BN* bn) {
                bn->used_batch = bn->follows_flag ? c.training : true;
                launch(c, "bn_finalize_kernel", 0, 64.0 * bn->C, [&]() {
                    return p3d_bn_finalize(bn_params(bn), M, bn->used_batch, bn->used_batch && c.update_moving, 1e-3f, c.s);
                });
            };
            BnApplyArgs a;
            memset(&a, 0, sizeof(a));
            a.mode = mode; a.M = M; a.C = C;
            a.y1 = y1->p; a.ld1 = y1->ld; a.scale1 = bn1->scale; a.shift1 = bn1->shift;
            if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; }
            if (two) { a.scale2 = bn2->scale; a.shift2 = bn2->shift; }
            a.z = out->p; a.ldz = out->ld;
            if (dropout && c.training && c.drop > 0.f) { a.drop_rate = c.drop; a.drop_scale = 1.f / (1.f - c.drop); a.seed = c.seed; a.seed_dev = c.seed_dev; }
            {   // few statistics partials per channel (stage 2): every apply block folds its own channels' -- one launch, not two or three
                const bool b1 = bn1->follows_flag ? c.training : true, b2 = two ? (bn2->follows_flag ? c.training : true) : false;
                if (!c.dry && p3d_bn_fold_apply_ok(M, C, b1 ? bn1->nparts : 0, b2 ? bn2->nparts : 0, a.drop_scale)) {
                    bn1->used_batch = b1;
                    if (two) bn2->used_batch = b2;
                    const BnParams p1 = bn_params(bn1), p2 = two ? bn_params(bn2) : BnParams{};
                    launch(c, "bn_fold_apply_kernel", 0, tens * (y2 ? 3 : 2), [&]() {
                        return p3d_bn_fold_apply(a, p1, p2, b1, b2, c.update_moving ? 1 : 0, 1e-3f, c.s);
                    });
                    return;
                }
            }
            fin(bn1);
            if (two) fin(bn2);
            launch(c, kn_apply.c_str(), 0, tens * (y2 ? 3 : 2), [&]() { return p3d_bn_apply(a, c.s); });
        };
        {
            auto scope = [](const BN* bn) { const std::string& n = bn->gamma->name; return n.substr(0, n.rfind('/')); };
            op.dec_kind = "bn"; op.dec_name1 = scope(bn1); op.dec_name2 = two ? scope(bn2) : std::string(); op.dec_act = y1;
            op.gates = [=](hipStream_t st, const float* ones, float* o1, float* o2, float* scratch) {
                if (small) {
                    BnSmallArgs a;
                    memset(&a, 0, sizeof(a));
                    a.mode = mode; a.M = (int)M; a.C = C; a.y1 = y1->p; a.ld1 = y1->ld;
                    if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; }
                    a.bn1 = bn_params(bn1);
                    if (two) a.bn2 = bn_params(bn2);
                    a.batch1 = 0; a.batch2 = 0; a.eps = 1e-3f;
                    a.dz = ones; a.lddz = C; a.dy1 = o1; a.lddy1 = C; a.dy2 = o2; a.lddy2 = C; a.acc2 = 0;
                    a.dgamma1 = scratch; a.dbeta1 = scratch + C; a.dgamma2 = scratch + 2 * C; a.dbeta2 = scratch + 3 * C;
                    HIPCHECK(p3d_bn_small_bwd(a, st));
                    return;
                }
                BnBwdArgs a;
                memset(&a, 0, sizeof(a));
                a.mode = mode; a.M = M; a.C = C; a.dz = ones; a.lddz = C;
                a.y1 = y1->p; a.ld1 = y1->ld; a.scale1 = bn1->scale; a.shift1 = bn1->shift; a.mean1 = bn1->mean; a.invstd1 = bn1->invstd;
                a.gamma1 = bn1->gamma->p; a.batch1 = 0; a.dy1 = o1; a.lddy1 = C;
                if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; a.dy2 = o2; a.lddy2 = C; }
                if (two) {
                    a.scale2 = bn2->scale; a.shift2 = bn2->shift; a.mean2 = bn2->mean; a.invstd2 = bn2->invstd;
                    a.gamma2 = bn2->gamma->p; a.batch2 = 0;
                }
                HIPCHECK(p3d_bn_bwd_apply(a, st));       // (dropout sites: the gate alone -- the keep pattern is the tests' own input)
            };
        }
        op.bwd = [=](const Ctx& c) {
            if (fused_site && c.fuse_bwd) return;
            if (small) {
                BnSmallArgs sa = small_args(c);
                sa.batch1 = bn1->used_batch; sa.batch2 = two ? bn2->used_batch : 0;
                launch(c, kn_sb.c_str(), 0, tens * (y2 ? 5 : 3), [&]() { return p3d_bn_small_bwd(sa, c.s); });
                return;
            }
            BnBwdArgs a;
            memset(&a, 0, sizeof(a));
            a.mode = mode; a.M = M; a.C = C;
            a.dz = out->g; a.lddz = out->ld;
            a.y1 = y1->p; a.ld1 = y1->ld; a.scale1 = bn1->scale; a.shift1 = bn1->shift; a.mean1 = bn1->mean; a.invstd1 = bn1->invstd;
            a.gamma1 = bn1->gamma->p; a.dgamma1 = bn1->gamma->g; a.dbeta1 = bn1->beta->g; a.batch1 = bn1->used_batch;
            a.part1 = statpart_arena + red_off; a.nparts = bwd_parts; a.coef1 = bnbuf + coef_off;
            a.dy1 = y1->g; a.lddy1 = y1->ld; a.acc1 = 0;
            if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; a.dy2 = y2->g; a.lddy2 = y2->ld; a.acc2 = *f2; }
            if (two) {
                a.scale2 = bn2->scale; a.shift2 = bn2->shift; a.mean2 = bn2->mean; a.invstd2 = bn2->invstd;
                a.gamma2 = bn2->gamma->p; a.dgamma2 = bn2->gamma->g; a.dbeta2 = bn2->beta->g; a.batch2 = bn2->used_batch;
                a.part2 = statpart_arena + red_off + (int64_t)bwd_parts * 2 * C; a.coef2 = bnbuf + coef_off + 2 * C;
            }
            if (dropout && c.training && c.drop > 0.f) { a.drop_rate = c.drop; a.drop_scale = 1.f / (1.f - c.drop); a.seed = c.seed; a.seed_dev = c.seed_dev; }
            launch(c, kn_red.c_str(), 0, tens * (y2 ? 3 : 2), [&]() { return p3d_bn_bwd_reduce(a, c.s); });
            launch(c, "bn_bwd_finalize_kernel", 0, 64.0 * C, [&]() { return p3d_bn_bwd_finalize(a, c.s); });
            launch(c, kn_bapply.c_str(), 0, tens * (y2 ? 5 : 3), [&]() { return p3d_bn_bwd_apply(a, c.s); });
        };
        ops.push_back(op);
        return out;
    }

    // tf.nn.max_pool3d SAME
    Act* maxpool(const std::string& opname, Act* x, const int k[3], const int s[3], Act* out, const std::string& out_name) {
        const ConvGeo g = make_geo(x->D, x->H, x->W, k, s);
        if (!out) out = new_act(out_name, x->N, g.O[0], g.O[1], g.O[2], x->C);
        else if (!out_name.empty()) named[out_name] = out;
        char* xflag = consume(x);
        // overlapping windows (pool1): the forward keeps the arg-max tap of every output so that the backward can gather
        unsigned* idx = nullptr;
        {
            const bool disjoint = g.k[0] == g.s[0] && g.k[1] == g.s[1] && g.k[2] == g.s[2] && !g.pad[0] && !g.pad[1] && !g.pad[2] &&
                                  g.O[0] * g.s[0] == g.I[0] && g.O[1] * g.s[1] == g.I[1] && g.O[2] * g.s[2] == g.I[2];
            if (!disjoint) idx = (unsigned*)dalloc<float>(out->rows() * (x->C / 4));
        }
        Op op;
        op.name = opname; op.kind = "maxpool";
        op.bytes = 4.0 * (x->rows() + out->rows()) * x->C;
        op.bbytes = 4.0 * (2.0 * x->rows() + 2.0 * out->rows()) * x->C;
        auto mk = [=]() {
            PoolArgs a;
            memset(&a, 0, sizeof(a));
            a.x = x->p; a.N = x->N; a.Di = x->D; a.Hi = x->H; a.Wi = x->W; a.C = x->C; a.ldx = x->ld;
            a.y = out->p; a.Do = g.O[0]; a.Ho = g.O[1]; a.Wo = g.O[2]; a.ldy = out->ld;
            // NB: k / s are pointers into the builder's stack; only the by-value geometry is safe here
            a.kd = g.k[0]; a.kh = g.k[1]; a.kw = g.k[2]; a.sd = g.s[0]; a.sh = g.s[1]; a.sw = g.s[2];
            a.pd = g.pad[0]; a.ph = g.pad[1]; a.pw = g.pad[2];
            a.dy = out->g; a.lddy = out->ld; a.dx = x->g; a.lddx = x->ld;
            a.idx = idx;
            return a;
        };
        const double pool_bytes = op.bytes;
        op.dec_kind = "pool"; op.dec_act = x;
        op.fwd = [=](const Ctx& c) { launch(c, "maxpool_fwd_kernel", 0, pool_bytes, [&]() { return p3d_maxpool_fwd(mk(), c.s); }); };
        op.bwd = [=](const Ctx& c) {
            const PoolArgs pa = mk();
            if (p3d_maxpool_disjoint(pa)) {
                launch(c, "maxpool_bwd_disjoint_kernel", 0, pool_bytes * 2, [&]() { return p3d_maxpool_bwd_disjoint(pa, *xflag, c.s); });
                return;
            }
            if (pa.idx) {
                launch(c, "maxpool_bwd_gather_kernel", 0, pool_bytes * 2, [&]() { return p3d_maxpool_bwd_gather(pa, *xflag, c.s); });
                return;
            }
            throw P3dError("max-pool with overlapping windows was built without its arg-max table");
        };
        ops.push_back(op);
        return out;
    }

    // ---- GroupNorm / CBAM variant (gn/p3d_gn.py) ---------------------------------------------------
    GN* add_gn(int C) {
        gns.emplace_back();
        GN* g = &gns.back();
        g->name = unique("group_norm");
        g->C = C; g->G = C < 32 ? C : 32; g->N = cfg.batch;
        g->gamma = add_param(g->name + "/gamma", {C}, true, INIT_ONES);
        g->beta = add_param(g->name + "/beta", {C}, true, INIT_ZEROS);
        const int64_t nc = (int64_t)cfg.batch * C;
        g->sums_off = stats_count; stats_count += 2 * nc;
        g->bsums_off = red_count; red_count += 2 * nc;
        g->tab_off = bnbuf_count; bnbuf_count += 7 * nc;
        return g;
    }
    GnParams gn_params(GN* g, bool bwd) {
        GnParams p;
        const int64_t nc = (int64_t)g->N * g->C;
        p.gamma = g->gamma->p; p.beta = g->beta->p;
        p.sums = bwd ? red_arena + g->bsums_off : stats_arena + g->sums_off;
        float* t = bnbuf + g->tab_off;
        p.scale = t; p.shift = t + nc; p.mean = t + 2 * nc; p.invstd = t + 3 * nc; p.coef = t + 4 * nc;
        p.C = g->C; p.G = g->G;
        return p;
    }

    // GroupNorm + fused activation pass (modes in gn.hip).  cb != null: mode 6, y2 is the CBAM input.
    Act* gn_apply(const std::string& opname, int mode, Act* y1, GN* g1, Act* y2, GN* g2, Act* out, const std::string& out_name,
                  bool dropout = false, CbamSite* cb = nullptr) {
        if (!out) out = new_act(out_name, y1->N, y1->D, y1->H, y1->W, y1->C);
        else if (!out_name.empty()) named[out_name] = out;
        consume(y1);
        char* f2 = nullptr;
        if (y2 && mode != 6) f2 = consume(y2);
        const int64_t M = y1->rows();
        const int R = y1->D * y1->H * y1->W, C = y1->C, N = y1->N;
        const double tens = (double)M * C * 4.0;
        const std::string ka = "gn_apply_kernel<" + std::to_string(mode) + ">";
        const std::string kr = "gn_bwd_reduce_kernel<" + std::to_string(mode) + ">";
        const std::string kb = "gn_bwd_apply_kernel<" + std::to_string(mode) + ">";
        Op op;
        op.name = opname; op.kind = "gn_apply" + std::to_string(mode);
        op.bytes = tens * (y2 ? 4 : 3); op.bbytes = tens * (y2 ? 7 : 5);
        op.owns = {g1->gamma, g1->beta};
        if (g2) { op.owns.push_back(g2->gamma); op.owns.push_back(g2->beta); }
        auto mk = [=](const Ctx& c, bool bwd) {
            GnApplyArgs a;
            memset(&a, 0, sizeof(a));
            a.mode = mode; a.M = M; a.R = R; a.C = C;
            a.y1 = y1->p; a.ld1 = y1->ld; a.g1 = gn_params(g1, bwd);
            if (y2) { a.y2 = y2->p; a.ld2 = y2->ld; }
            if (g2) a.g2 = gn_params(g2, bwd);
            if (cb) { a.cs = bnbuf + cb->buf_off + cbam_cs_off(cb); a.ss = bnbuf + cb->buf_off + cbam_ss_off(cb); }
            a.z = out->p; a.ldz = out->ld; a.dz = out->g;
            a.dy1 = y1->g; a.lddy1 = y1->ld;
            if (mode == 6) { a.dy2 = cb->dout; a.lddy2 = C; a.acc2 = 0; }
            else if (y2) { a.dy2 = y2->g; a.lddy2 = y2->ld; a.acc2 = *f2; }
            if (dropout && c.training && c.drop > 0.f) { a.drop_rate = c.drop; a.drop_scale = 1.f / (1.f - c.drop); a.seed = c.seed; a.seed_dev = c.seed_dev; }
            a.eps = 1e-5f;
            a.dgamma1 = g1->gamma->g; a.dbeta1 = g1->beta->g;
            if (g2) { a.dgamma2 = g2->gamma->g; a.dbeta2 = g2->beta->g; }
            return a;
        };
        // small tensors (every GroupNorm of stage 3): one launch each way, see gn.hip
        static const bool no_small = p3d_tune_env("P3D_NO_GN_SMALL") != nullptr;
        const bool small = !no_small && !dropout && p3d_gn_small_ok(R, C, g1->G) && (!g2 || g2->G == g1->G);
        const std::string ksf = "gn_small_fwd_kernel<" + std::to_string(mode) + ">";
        const std::string ksb = "gn_small_bwd_kernel<" + std::to_string(mode) + ">";
        op.fwd = [=](const Ctx& c) {
            const GnApplyArgs a = mk(c, false);
            if (small) {
                launch(c, ksf.c_str(), 0, tens * (y2 ? 3 : 2), [&]() { return p3d_gn_small_fwd(a, c.s); });
                return;
            }
            launch(c, "gn_stats_kernel", 0, tens, [&]() { return p3d_gn_stats(y1->p, y1->ld, N, R, C, a.g1.sums, c.s); });
            launch(c, "gn_finalize_kernel", 0, 32.0 * N * C, [&]() { return p3d_gn_finalize(a.g1, N, R, 1e-5f, c.s); });
            if (mode == 3) {
                launch(c, "gn_stats_kernel", 0, tens, [&]() { return p3d_gn_stats(y2->p, y2->ld, N, R, C, a.g2.sums, c.s); });
                launch(c, "gn_finalize_kernel", 0, 32.0 * N * C, [&]() { return p3d_gn_finalize(a.g2, N, R, 1e-5f, c.s); });
            }
            launch(c, ka.c_str(), 0, tens * (y2 ? 3 : 2), [&]() { return p3d_gn_apply(a, c.s); });
        };
        op.bwd = [=](const Ctx& c) {
            const GnApplyArgs a = mk(c, true);
            if (small) {
                launch(c, ksb.c_str(), 0, tens * (y2 ? 5 : 3), [&]() { return p3d_gn_small_bwd(a, c.s); });
                return;
            }
            launch(c, kr.c_str(), 0, tens * (y2 ? 3 : 2), [&]() { return p3d_gn_bwd_reduce(a, c.s); });
            launch(c, "gn_bwd_finalize_kernel", 0, 64.0 * N * C, [&]() { return p3d_gn_bwd_finalize(a.g1, N, R, g1->gamma->g, g1->beta->g, c.s); });
            if (mode == 3)
                launch(c, "gn_bwd_finalize_kernel", 0, 64.0 * N * C, [&]() { return p3d_gn_bwd_finalize(a.g2, N, R, g2->gamma->g, g2->beta->g, c.s); });
            launch(c, kb.c_str(), 0, tens * (y2 ? 5 : 3), [&]() { return p3d_gn_bwd_apply(a, c.s); });
        };
        ops.push_back(op);
        return out;
    }

    // scratch layout of one CBAM site inside bnbuf (floats)
    static int64_t cbam_part_off(CbamSite*) { return 0; }
    int64_t cbam_vec_off(CbamSite* cb) { return (int64_t)cb->x->N * cb->chunks * cb->x->C * 3; }                // avg, mx, ties, cs, davg, dmx: [N][C] each
    int64_t cbam_cs_off(CbamSite* cb) { return cbam_vec_off(cb) + 3 * (int64_t)cb->x->N * cb->x->C; }
    int64_t cbam_h_off(CbamSite* cb) { return cbam_vec_off(cb) + 6 * (int64_t)cb->x->N * cb->x->C; }            // havg, hmx [N][C/8]
    int64_t cbam_sp_off(CbamSite* cb) { return cbam_h_off(cb) + 4 * (int64_t)cb->x->N * (cb->x->C / 8) + 8; }   // (+ dh [N][2][C/8]) then sp [M][2]
    int64_t cbam_ss_off(CbamSite* cb) { return cbam_sp_off(cb) + 2 * cb->x->rows(); }                          // ss [M]
    int64_t cbam_dpre_off(CbamSite* cb) { return cbam_ss_off(cb) + cb->x->rows(); }                             // dpre [M]
    int64_t cbam_dsp_off(CbamSite* cb) { return cbam_dpre_off(cb) + cb->x->rows(); }                            // dsp [M][2]
    int64_t cbam_dcs_off(CbamSite* cb) { return cbam_dsp_off(cb) + 2 * cb->x->rows(); }                         // dcs_part [N][chunks][C]
    int64_t cbam_dO_off(CbamSite* cb) { return cbam_dcs_off(cb) + (int64_t)cb->x->N * cb->chunks * cb->x->C; }   // dO [N][C]
    int64_t cbam_total(CbamSite* cb) { return ((cbam_dO_off(cb) + (int64_t)cb->x->N * cb->x->C + 63) / 64) * 64; }

    CbamArgs cbam_args(CbamSite* cb) {
        CbamArgs a;
        memset(&a, 0, sizeof(a));
        Act* x = cb->x;
        const int64_t nc = (int64_t)x->N * x->C;
        float* b = bnbuf + cb->buf_off;
        a.x = x->p; a.ld = x->ld; a.N = x->N; a.D = x->D; a.H = x->H; a.W = x->W; a.C = x->C; a.Ch = x->C / 8;
        a.k0 = cb->k0->p; a.b0 = cb->b0->p; a.k1 = cb->k1->p; a.b1 = cb->b1->p; a.k7 = cb->k7->p;
        a.chunks = cb->chunks;
        a.part = b + cbam_part_off(cb);
        float* v = b + cbam_vec_off(cb);
        a.avg = v; a.mx = v + nc; a.ties = v + 2 * nc; a.cs = v + 3 * nc; a.davg = v + 4 * nc; a.dmx = v + 5 * nc;
        a.havg = b + cbam_h_off(cb); a.hmx = a.havg + (int64_t)x->N * a.Ch; a.dh = a.hmx + (int64_t)x->N * a.Ch;
        a.sp = b + cbam_sp_off(cb); a.ss = b + cbam_ss_off(cb); a.dpre = b + cbam_dpre_off(cb); a.dsp = b + cbam_dsp_off(cb);
        a.dcs_part = b + cbam_dcs_off(cb);
        a.dO = b + cbam_dO_off(cb);
        a.dout = cb->dout;
        a.dx = x->g; a.lddx = x->ld; a.accx = cb->xflag ? *cb->xflag : 0;
        a.dk0 = cb->k0->g; a.db0 = cb->b0->g; a.dk1 = cb->k1->g; a.db1 = cb->b1->g; a.dk7 = cb->k7->g;
        return a;
    }

    // residual = cbam_block(residual, 'cbam_<id>')  (gn/p3d_gn.py:175)
    CbamSite* cbam(Act* x, int id) {
        cbams.emplace_back();
        CbamSite* cb = &cbams.back();
        const std::string nm = "cbam_" + std::to_string(id);
        const int C = x->C, Ch = C / 8;
        if (Ch < 1) throw P3dError("CBAM needs at least 8 channels");
        cb->k0 = add_param(nm + "/ch_at/mlp_0/kernel", {C, Ch}, true, INIT_VS);
        cb->b0 = add_param(nm + "/ch_at/mlp_0/bias", {Ch}, true, INIT_ZEROS);
        cb->k1 = add_param(nm + "/ch_at/mlp_1/kernel", {Ch, C}, true, INIT_VS);
        cb->b1 = add_param(nm + "/ch_at/mlp_1/bias", {C}, true, INIT_ZEROS);
        cb->k7 = add_param(nm + "/sp_at/conv3d/kernel", {7, 7, 7, 2, 1}, true, INIT_VS);
        cb->x = x;
        const int R = x->D * x->H * x->W;
        int chunks = R / 16;
        if (chunks < 1) chunks = 1;
        if (chunks > 64) chunks = 64;
        cb->chunks = chunks;
        cb->dout = dalloc<float>(x->rows() * C);
        cb->buf_off = bnbuf_count; bnbuf_count += cbam_total(cb);
        cb->xflag = consume(x);
        Op op;
        op.name = "block" + std::to_string(id) + "/cbam"; op.kind = "cbam";
        op.bytes = 4.0 * 2 * x->rows() * C; op.bbytes = 4.0 * 7 * x->rows() * C;
        op.owns = {cb->k0, cb->b0, cb->k1, cb->b1, cb->k7};
        op.fwd = [=](const Ctx& c) { launch(c, "cbam_forward(5 kernels)", 0, 8.0 * x->rows() * C, [&]() { return p3d_cbam_forward(cbam_args(cb), c.s); }); };
        op.bwd = [=](const Ctx& c) { launch(c, "cbam_backward(6 kernels)", 0, 28.0 * x->rows() * C, [&]() { return p3d_cbam_backward(cbam_args(cb), c.s); }); };
        ops.push_back(op);
        return cb;
    }

    // Bottleneck.infer of the GN file, gn/p3d_gn.py:127-179
    Act* bottleneck_gn(Act* x, int id, int inplanes, int planes, bool first, bool stride2) {
        blocks[id].in = x; blocks[id].op0 = ops.size();
        Act* r = bottleneck_gn_body(x, id, inplanes, planes, first, stride2);
        blocks[id].out = r; blocks[id].op1 = ops.size();
        return r;
    }
    Act* bottleneck_gn_body(Act* x, int id, int inplanes, int planes, bool first, bool stride2) {
        const std::string sid = std::to_string(id);
        const char st = "ABC"[id % 3];
        const int one[3] = {1, 1, 1}, s2[3] = {1, 2, 2};
        const int* s = (first && stride2) ? s2 : one;
        const int kS[3] = {1, 3, 3}, kT[3] = {3, 1, 1};
        const std::string B = "block" + sid + "/";
        Param* w1 = conv_weight("conv3_" + sid + "_1", {1, 1, 1, inplanes, planes});
        GN* g1 = add_gn(planes);
        Act* y1 = conv(B + "conv1", x, w1, nullptr, one, s, planes, nullptr, "");
        Act* z1 = gn_apply(B + "gn1", 0, y1, g1, nullptr, nullptr, nullptr, B + "conv1_bn_relu");
        const std::string nm = std::string("ST") + st + "_" + sid + "_2";
        Param* wS = conv_weight(nm + "_S", {1, 3, 3, planes, planes});
        Param* bS = conv_weight(nm + "_S_bias", {planes});
        GN* gS = add_gn(planes);
        Act* yS = conv(B + "convS", z1, wS, bS, kS, one, planes, nullptr, "");
        Act* zS = (st == 'B') ? nullptr : gn_apply(B + "gnS", 0, yS, gS, nullptr, nullptr, nullptr, "");
        Param* wT = conv_weight(nm + "_T", {3, 1, 1, planes, planes});
        Param* bT = conv_weight(nm + "_T_bias", {planes});
        GN* gT = add_gn(planes);
        Act* yT = conv(B + "convT", st == 'B' ? z1 : zS, wT, bT, kT, one, planes, nullptr, "");
        Act* stout;
        if (st == 'A') stout = gn_apply(B + "gnT", 0, yT, gT, nullptr, nullptr, nullptr, B + "st");
        else if (st == 'B') stout = gn_apply(B + "gnST", 3, yT, gT, yS, gS, nullptr, B + "st");      // relu(gn(T)) + relu(gn(S))
        else stout = gn_apply(B + "gnT", 4, yT, gT, zS, nullptr, nullptr, B + "st");
        Param* w3 = conv_weight("conv3_" + sid + "_3", {1, 1, 1, planes, planes * 4});
        GN* g3 = add_gn(planes * 4);
        Act* y3 = conv(B + "conv3", stout, w3, nullptr, one, one, planes * 4, nullptr, "");
        Act* res = x;
        if (first) {
            Param* wp = conv_weight("dw3d_" + sid, {1, 1, 1, inplanes, planes * 4});
            GN* gp = add_gn(planes * 4);
            Act* yp = conv(B + "proj", x, wp, nullptr, one, s, planes * 4, nullptr, "");
            res = gn_apply(B + "gnp", 5, yp, gp, nullptr, nullptr, nullptr, "");
        }
        CbamSite* cb = cbam(res, id);
        return gn_apply(B + "gn3", 6, y3, g3, res, nullptr, nullptr, B + "out", false, cb);
    }

    // inference_p3d, gn/p3d_gn.py:214-258
    // Stem + three GN/CBAM stages + temporal pools shared by the heads of gn/p3d_gn.py (:215-240, :491-516).
    // before_stage(stage, input) runs before the blocks of stages 1 and 2 so that a head can create its skip
    // transposed convs where the reference does (GroupNorm scopes are numbered in creation order).
    void build_gn_encoder(Act* skip2, const std::function<void(int, Act*)>& before_stage, Act* pools[3]) {
        const int B = cfg.batch, T = cfg.frames, H = cfg.height, W = cfg.width, b = cfg.base;
        if (T % 16 || H % 16 || W % 16) throw P3dError("frames/height/width must be multiples of 16");
        if (b % 8) throw P3dError("base must be a multiple of 8");
        x_in = new_act("x", B, T, H, W, 3, false);
        const int k177[3] = {1, 7, 7}, s122[3] = {1, 2, 2}, k233[3] = {2, 3, 3}, s222[3] = {2, 2, 2};
        const int k211[3] = {2, 1, 1}, s211[3] = {2, 1, 1};
        Param* w0 = conv_weight("firstconv1", {1, 7, 7, 3, b});
        GN* g0 = add_gn(b);
        Act* c1 = conv("stem/conv", x_in, w0, nullptr, k177, s122, b, nullptr, "conv1_custom", true);
        Act* a1 = gn_apply("stem/gn", 0, c1, g0, nullptr, nullptr, nullptr, "conv1_custom_bn_relu");
        Act* cur = maxpool("pool1", a1, k233, s222, nullptr, "pool1");
        int id = 0, inpl = b;
        const int planes[3] = {b, 2 * b, 4 * b};
        Act* skips[3] = {skip2, nullptr, nullptr};
        const char* pool_names[3] = {"pool2", "pool3", "pool4"};
        for (int stage = 0; stage < 3; ++stage) {
            if (stage > 0 && before_stage) before_stage(stage, cur);
            for (int j = 0; j < cfg.blocks[stage]; ++j) {
                cur = bottleneck_gn(cur, id, inpl, planes[stage], j == 0, stage > 0);
                inpl = planes[stage] * 4;
                ++id;
            }
            cur = maxpool(pool_names[stage], cur, k211, s211, skips[stage], pool_names[stage]);
            pools[stage] = cur;
        }
    }
    // tf.layers.conv3d / conv3d_transpose + GNReLU (gn/p3d_gn.py:14-22,49-51), optionally into a concat slice
    Act* gn_layer(const char* name, bool up, Act* x, int filters, const int* k, const int* s, Act* out,
                  const std::string& out_name, bool dropout = false) {
        Param* kern = up ? conv_weight(std::string(name) + "/kernel", {k[0], k[1], k[2], filters, x->C})
                         : conv_weight(std::string(name) + "/kernel", {k[0], k[1], k[2], x->C, filters});
        Param* bi = add_param(std::string(name) + "/bias", {filters}, true, INIT_ZEROS);
        GN* g = add_gn(filters);
        Act* y = up ? deconv(name, x, kern, bi, k, s, filters, nullptr, "") : conv(name, x, kern, bi, k, s, filters, nullptr, "");
        return gn_apply(std::string(name) + "_gn", 0, y, g, nullptr, nullptr, out, out_name, dropout);
    }

    // inference_p3d (gn/p3d_gn.py:214-258; pool4_filters = 16) and inference_p3d_concat (gn/p3d_gn.py:279-324;
    // pool4_filters = 8, the only difference), filters in units of base
    void build_gn_p3d(int pool4_filters) {
        const int B = cfg.batch, T = cfg.frames, H = cfg.height, W = cfg.width, b = cfg.base;
        const int k333[3] = {3, 3, 3}, s111[3] = {1, 1, 1}, s222[3] = {2, 2, 2}, s444[3] = {4, 4, 4};
        // concatenator = [deconv_pool3_gn (8b) | deconv_pool4_gn (16b or 8b) | pool2 (4b)]  (gn/p3d_gn.py:251,317)
        const int p4 = pool4_filters * b;
        Act* cat = new_act("concatenator", B, T / 4, H / 4, W / 4, 12 * b + p4);
        Act* pools[3] = {nullptr, nullptr, nullptr};
        build_gn_encoder(new_view(cat, 8 * b + p4, 4 * b, "pool2"), [&](int stage, Act* in) {
            if (stage == 2) gn_layer("deconv_pool3", true, in, 8 * b, k333, s222, new_view(cat, 0, 8 * b, ""), "");   // before stage 3
        }, pools);
        gn_layer("deconv_pool4", true, pools[2], p4, k333, s444, new_view(cat, 8 * b, p4, ""), "");
        Act* zc = gn_layer("conv_concat", false, cat, 16 * b, k333, s111, nullptr, "conv_concat");
        Act* zr = gn_layer("deconv_revise", true, zc, 4 * b, k333, s222, nullptr, "deconv_revise", /*dropout=*/true);
        Param* kp = conv_weight("predict_revise/kernel", {3, 3, 3, 1, 4 * b});
        Param* bp = add_param("predict_revise/bias", {1}, true, INIT_ZEROS);
        head(zr, kp, bp, /*with_sigmoid=*/false);
    }

    // inference_p3d_decoder_block (gn/p3d_gn.py:489-539, net = 'P3D_DECODER' in gn/train_p3d_gn_dataset.py:177):
    // everything lives in tf.variable_scope('P3D'); skip deconvs of pool2/3/4 to 4x28x28, concat, conv_concat, two
    // conv-deconv-conv decoder blocks narrowing to base/4 channels at full resolution, dropout, and a plain
    // 3x3x3 conv to one channel (no sigmoid).
    void build_gn_decoder() {
        const int B = cfg.batch, T = cfg.frames, H = cfg.height, W = cfg.width, b = cfg.base;
        if (b % 16) throw P3dError("the decoder-block head needs base to be a multiple of 16");
        var_prefix = "P3D/";
        const int k333[3] = {3, 3, 3}, k233[3] = {2, 3, 3}, k133[3] = {1, 3, 3};
        const int s111[3] = {1, 1, 1}, s222[3] = {2, 2, 2}, s444[3] = {4, 4, 4};
        Act* cat = new_act("concatenator", B, T / 4, H / 4, W / 4, 14 * b);   // [deconv_pool2 2b | deconv_pool3 4b | deconv_pool4 8b]
        Act* pools[3] = {nullptr, nullptr, nullptr};
        build_gn_encoder(nullptr, [&](int stage, Act* in) {
            if (stage == 1) gn_layer("deconv_pool2", true, in, 2 * b, k333, s111, new_view(cat, 0, 2 * b, ""), "deconv_pool2");
            else gn_layer("deconv_pool3", true, in, 4 * b, k233, s222, new_view(cat, 2 * b, 4 * b, ""), "deconv_pool3");
        }, pools);
        gn_layer("deconv_pool4", true, pools[2], 8 * b, k133, s444, new_view(cat, 6 * b, 8 * b, ""), "deconv_pool4");
        Act* z = gn_layer("conv_concat", false, cat, 16 * b, k333, s111, nullptr, "conv_concat");
        z = gn_layer("decoder1_conv1", false, z, 4 * b, k333, s111, nullptr, "decoder1_conv1");
        z = gn_layer("decoder1_deconv", true, z, 4 * b, k333, s222, nullptr, "decoder1_deconv");
        z = gn_layer("decoder1_conv2", false, z, 2 * b, k333, s111, nullptr, "decoder1_conv2");
        z = gn_layer("decoder2_conv1", false, z, b / 2, k333, s111, nullptr, "decoder2_conv1");
        z = gn_layer("decoder2_deconv", true, z, b / 2, k333, s222, nullptr, "decoder2_deconv");
        z = gn_layer("decoder2_conv2", false, z, b / 4, k333, s111, nullptr, "decoder2_conv2", /*dropout=*/true);
        Param* kp = conv_weight("results/kernel", {3, 3, 3, b / 4, 1});
        Param* bp = add_param("results/bias", {1}, true, INIT_ZEROS);
        head(z, kp, bp, /*with_sigmoid=*/false, /*transpose=*/false);
    }

    // ---- the reference graph -------------------------------------------------------------------
    Param* conv_weight(const std::string& name, std::vector<int64_t> shape) { return add_param(name, shape, true, INIT_XAVIER); }

    // Bottleneck.infer, p3d.py:83-136 (3-D branch only; the 2-D branch is unreachable)
    // per bottleneck: input / output tensors and the [first, last) range of its ops, for p3d_block_forward
    struct BlockInfo { Act* in = nullptr; Act* out = nullptr; size_t op0 = 0, op1 = 0; };
    std::map<int, BlockInfo> blocks;
    Act* bottleneck(Act* x, int id, int inplanes, int planes, bool first, bool stride2) {
        blocks[id].in = x; blocks[id].op0 = ops.size();
        Act* r = bottleneck_body(x, id, inplanes, planes, first, stride2);
        blocks[id].out = r; blocks[id].op1 = ops.size();
        return r;
    }
    Act* bottleneck_body(Act* x, int id, int inplanes, int planes, bool first, bool stride2) {
        const std::string sid = std::to_string(id);
        const char st = "ABC"[id % 3];
        const int one[3] = {1, 1, 1};
        const int s2[3] = {1, 2, 2};
        const int* s = (first && stride2) ? s2 : one;
        const int kS[3] = {1, 3, 3}, kT[3] = {3, 1, 1};
        const std::string B = "block" + sid + "/";
        // variable creation order matters for BN auto-naming
        // BatchNorm fusion (ConvFuse): bn1, bnS, bnT and their ReLUs ride on the operand paths of the convs around them;
        // only bn3 (+ residual) keeps a pass of its own.  Every bn_apply below marked fused_site runs only with fusion off.
        Param* w1 = conv_weight("conv3_" + sid + "_1", {1, 1, 1, inplanes, planes});
        BN* bn1 = add_bn("", planes, false);
        const ConvGeo g1 = make_geo(x->D, x->H, x->W, one, s);
        const bool fz = (int64_t)x->N * g1.O[0] * g1.O[1] * g1.O[2] <= fuse_max_rows;      // this bottleneck is built fusable
        ConvFuse f1; if (fz) f1.out_bn = bn1;
        Act* y1 = conv(B + "conv1", x, w1, nullptr, one, s, planes, bn1, "", false, false, 0, &f1);
        Act* z1 = bn_apply(B + "bn1", 0, y1, bn1, nullptr, nullptr, nullptr, B + "conv1_bn_relu", false, /*fused_site=*/fz);
        const std::string nm = std::string("ST") + st + "_" + sid + "_2";
        Act* stout = nullptr;
        ConvFuse f3;                 // conv3's view of the ST output
        if (st == 'A') {          // p3d.py:56-63
            Param* wS = conv_weight(nm + "_S", {1, 3, 3, planes, planes});
            Param* bS = conv_weight(nm + "_S_bias", {planes});
            BN* bnS = add_bn("", planes, false);
            ConvFuse fS; fS.at = P3D_AT_RELU1; fS.src[0] = {y1, bn1, 1}; fS.ngate = 1; fS.gate[0] = {y1, bn1, 0}; fS.out_bn = bnS;
            if (!fz) fS = ConvFuse();
            Act* yS = conv(B + "convS", z1, wS, bS, kS, one, planes, bnS, "", false, false, 0, &fS);
            Act* zS = bn_apply(B + "bnS", 0, yS, bnS, nullptr, nullptr, nullptr, "", false, fz);
            Param* wT = conv_weight(nm + "_T", {3, 1, 1, planes, planes});
            Param* bT = conv_weight(nm + "_T_bias", {planes});
            BN* bnT = add_bn("", planes, false);
            ConvFuse fT; fT.at = P3D_AT_RELU1; fT.src[0] = {yS, bnS, 1}; fT.ngate = 1; fT.gate[0] = {yS, bnS, 0}; fT.out_bn = bnT;
            if (!fz) fT = ConvFuse();
            Act* yT = conv(B + "convT", zS, wT, bT, kT, one, planes, bnT, "", false, false, 0, &fT);
            stout = bn_apply(B + "bnT", 0, yT, bnT, nullptr, nullptr, nullptr, B + "st", false, fz);
            f3.at = P3D_AT_RELU1; f3.src[0] = {yT, bnT, 1}; f3.ngate = 1; f3.gate[0] = {yT, bnT, 0};
        } else if (st == 'B') {   // p3d.py:65-72
            Param* wS = conv_weight(nm + "_S", {1, 3, 3, planes, planes});
            Param* bS = conv_weight(nm + "_S_bias", {planes});
            BN* bnS = add_bn("", planes, false);
            // z1 feeds both siblings: convS (registered first, so last in backward) folds and publishes bn1 in the forward and,
            // in the backward, adds the raw gradient convT left in z1->g, gates it and reduces for bn1
            ConvFuse fS; fS.at = P3D_AT_RELU1; fS.src[0] = {y1, bn1, 1}; fS.ngate = 1; fS.gate[0] = {y1, bn1, 0}; fS.accum_in = true; fS.out_bn = bnS;
            if (!fz) fS = ConvFuse();
            Act* yS = conv(B + "convS", z1, wS, bS, kS, one, planes, bnS, "", false, false, /*sibling=*/1, &fS);
            Param* wT = conv_weight(nm + "_T", {3, 1, 1, planes, planes});
            Param* bT = conv_weight(nm + "_T_bias", {planes});
            BN* bnT = add_bn("", planes, false);
            ConvFuse fT; fT.at = P3D_AT_RELU1; fT.src[0] = {y1, bn1, 2}; fT.out_bn = bnT;
            if (!fz) fT = ConvFuse();
            Act* yT = conv(B + "convT", z1, wT, bT, kT, one, planes, bnT, "", false, false, /*sibling=*/2, &fT);
            stout = bn_apply(B + "bnST", 3, yS, bnS, yT, bnT, nullptr, B + "st", false, fz);
            f3.at = P3D_AT_RELU2; f3.src[0] = {yS, bnS, 1}; f3.src[1] = {yT, bnT, 1};
            f3.ngate = 2; f3.gate[0] = {yS, bnS, 0}; f3.gate[1] = {yT, bnT, 0};
        } else {                  // p3d.py:74-81
            Param* wS = conv_weight(nm + "_S", {1, 3, 3, planes, planes});
            Param* bS = conv_weight(nm + "_S_bias", {planes});
            BN* bnS = add_bn("", planes, false);
            ConvFuse fS; fS.at = P3D_AT_RELU1; fS.src[0] = {y1, bn1, 1}; fS.ngate = 1; fS.gate[0] = {y1, bn1, 0}; fS.out_bn = bnS;
            if (!fz) fS = ConvFuse();
            Act* yS = conv(B + "convS", z1, wS, bS, kS, one, planes, bnS, "", false, false, 0, &fS);
            Act* zS = bn_apply(B + "bnS", 0, yS, bnS, nullptr, nullptr, nullptr, "", false, fz);
            Param* wT = conv_weight(nm + "_T", {3, 1, 1, planes, planes});
            Param* bT = conv_weight(nm + "_T_bias", {planes});
            BN* bnT = add_bn("", planes, false);
            // zS also reaches conv3 through the skip: conv3's input gradient leaves its raw result in zS->g, convT's adds its
            // own, gates and reduces for bnS
            ConvFuse fT; fT.at = P3D_AT_RELU1; fT.src[0] = {yS, bnS, 1}; fT.ngate = 1; fT.gate[0] = {yS, bnS, 0}; fT.accum_in = true; fT.out_bn = bnT;
            if (!fz) fT = ConvFuse();
            Act* yT = conv(B + "convT", zS, wT, bT, kT, one, planes, bnT, "", false, false, 0, &fT);
            stout = bn_apply(B + "bnT", 4, yT, bnT, zS, nullptr, nullptr, B + "st", false, fz);
            f3.at = P3D_AT_RELU2; f3.src[0] = {yS, bnS, 0}; f3.src[1] = {yT, bnT, 1};
            f3.ngate = 1; f3.gate[0] = {yT, bnT, 0}; f3.raw = zS; f3.raw_flag = zS->last_flag;      // the flag of bnT's skip read
        }
        if (!fz) f3 = ConvFuse();
        Param* w3 = conv_weight("conv3_" + sid + "_3", {1, 1, 1, planes, planes * 4});
        BN* bn3 = add_bn("", planes * 4, false);
        Act* y3 = conv(B + "conv3", stout, w3, nullptr, one, one, planes * 4, bn3, "", false, false, 0, &f3);
        if (first) {              // p3d.py:124-127
            Param* wp = conv_weight("dw3d_" + sid, {1, 1, 1, inplanes, planes * 4});
            BN* bnp = add_bn("", planes * 4, false);
            Act* yp = conv(B + "proj", x, wp, nullptr, one, s, planes * 4, bnp, "");
            return bn_apply(B + "bn3", 2, y3, bn3, yp, bnp, nullptr, B + "out");
        }
        return bn_apply(B + "bn3", 1, y3, bn3, x, nullptr, nullptr, B + "out");
    }

    // p3d.py:170-195: stem + three stages + temporal pools, shared verbatim by every head.  skip[0..1] are
    // where pool2 / pool3 land (channel slices of decoder concat buffers for the unet head, or null).
    Act* stem_out = nullptr;     // conv1_custom_bn_relu, which the unet++ head pools a second time (p3d.py:408)
    void build_encoder(Act* skip2, Act* skip3, Act*& pool2, Act*& pool3, Act*& pool4) {
        const int B = cfg.batch, T = cfg.frames, H = cfg.height, W = cfg.width, b = cfg.base;
        if (T % 16 || H % 16 || W % 16) throw P3dError("frames/height/width must be multiples of 16");
        if (b % 8) throw P3dError("base must be a multiple of 8");
        x_in = new_act("x", B, T, H, W, 3, /*with_grad=*/false);
        // p3d.py:172-177
        const int k177[3] = {1, 7, 7}, s122[3] = {1, 2, 2};
        Param* w0 = conv_weight("firstconv1", {1, 7, 7, 3, b});
        BN* bn0 = add_bn("", b, true);
        Act* c1 = conv("stem/conv", x_in, w0, nullptr, k177, s122, b, bn0, "conv1_custom", /*stem=*/true);
        Act* a1 = bn_apply("stem/bn", 0, c1, bn0, nullptr, nullptr, nullptr, "conv1_custom_bn_relu");
        stem_out = a1;
        const int k233[3] = {2, 3, 3}, s222[3] = {2, 2, 2}, k211[3] = {2, 1, 1}, s211[3] = {2, 1, 1};
        Act* cur = maxpool("pool1", a1, k233, s222, nullptr, "pool1");
        int id = 0, inpl = b;
        const int planes[3] = {b, 2 * b, 4 * b};
        Act* skips[3] = {skip2, skip3, nullptr};
        Act* outs[3] = {nullptr, nullptr, nullptr};
        const char* pool_names[3] = {"pool2", "pool3", "pool4"};
        for (int stage = 0; stage < 3; ++stage) {
            for (int j = 0; j < cfg.blocks[stage]; ++j) {
                cur = bottleneck(cur, id, inpl, planes[stage], j == 0, stage > 0);
                inpl = planes[stage] * 4;
                ++id;
            }
            cur = maxpool(pool_names[stage], cur, k211, s211, skips[stage], pool_names[stage]);
            outs[stage] = cur;
        }
        pool2 = outs[0]; pool3 = outs[1]; pool4 = outs[2];
    }

    void build_unet() {
        const int B = cfg.batch, T = cfg.frames, H = cfg.height, W = cfg.width, b = cfg.base;
        const int k233[3] = {2, 3, 3}, s222[3] = {2, 2, 2};
        // concat buffers of the decoder (p3d.py:203,208): [deconvN_re | poolM]
        Act* cat1 = new_act("deconv1_concat", B, T / 8, H / 8, W / 8, 16 * b);
        Act* cat2 = new_act("deconv2_concat", B, T / 4, H / 4, W / 4, 8 * b);
        Act *pool2, *pool3, *pool4;
        build_encoder(new_view(cat2, 4 * b, 4 * b, "pool2"), new_view(cat1, 8 * b, 8 * b, "pool3"), pool2, pool3, pool4);
        // decoder p3d.py:200-219
        const int k133[3] = {1, 3, 3}, k333[3] = {3, 3, 3}, k111[3] = {1, 1, 1}, s111[3] = {1, 1, 1};
        {
            Param* k = conv_weight("conv3d_transpose/kernel", {1, 3, 3, 8 * b, 16 * b});
            Param* bi = add_param("conv3d_transpose/bias", {8 * b}, true, INIT_ZEROS);
            BN* bn = add_bn("deconv1_bn", 8 * b, true);
            Act* y = deconv("deconv1", pool4, k, bi, k133, s222, 8 * b, bn, "");
            bn_apply("deconv1_bn", 0, y, bn, nullptr, nullptr, new_view(cat1, 0, 8 * b, "deconv1_re"), "");
        }
        {
            Param* k = conv_weight("conv3d_transpose_1/kernel", {2, 3, 3, 4 * b, 16 * b});
            Param* bi = add_param("conv3d_transpose_1/bias", {4 * b}, true, INIT_ZEROS);
            BN* bn = add_bn("deconv2_bn", 4 * b, true);
            Act* y = deconv("deconv2", cat1, k, bi, k233, s222, 4 * b, bn, "");
            bn_apply("deconv2_bn", 0, y, bn, nullptr, nullptr, new_view(cat2, 0, 4 * b, "deconv2_re"), "");
        }
        Act* d3;
        {
            Param* k = conv_weight("conv3d_transpose_2/kernel", {3, 3, 3, 2 * b, 8 * b});
            Param* bi = add_param("conv3d_transpose_2/bias", {2 * b}, true, INIT_ZEROS);
            BN* bn = add_bn("deconv3_bn", 2 * b, true);
            Act* y = deconv("deconv3", cat2, k, bi, k333, s222, 2 * b, bn, "", /*bn_has_dropout=*/true);
            d3 = bn_apply("deconv3_bn", 0, y, bn, nullptr, nullptr, nullptr, "deconv3_re", /*dropout=*/true);
        }
        Param* k4 = conv_weight("conv3d/kernel", {1, 1, 1, 2 * b, b / 2});
        Param* b4 = add_param("conv3d/bias", {b / 2}, true, INIT_ZEROS);
        Act* c4 = conv("deconv4_conv1", d3, k4, b4, k111, s111, b / 2, nullptr, "deconv4_conv1");
        Param* k5 = conv_weight("conv3d_transpose_3/kernel", {3, 3, 3, 1, b / 2});
        Param* b5 = add_param("conv3d_transpose_3/bias", {1}, true, INIT_ZEROS);
        head(c4, k5, b5);
    }

    // p3d_concat (p3d.py:224-276, --structure concat): three skip deconvs to 4x28x28, channel concat, one dense
    // 3x3x3 conv, one deconv, and a 1-channel deconv WITHOUT sigmoid.
    void build_concat() {
        const int B = cfg.batch, T = cfg.frames, H = cfg.height, W = cfg.width, b = cfg.base;
        const int k333[3] = {3, 3, 3}, s111[3] = {1, 1, 1}, s222[3] = {2, 2, 2}, s444[3] = {4, 4, 4};
        Act *pool2, *pool3, *pool4;
        build_encoder(nullptr, nullptr, pool2, pool3, pool4);
        Act* cat = new_act("concatenator", B, T / 4, H / 4, W / 4, 14 * b);
        auto up = [&](const char* name, const char* bn_name, Act* x, int filters, const int* s, int coff) {
            Param* k = conv_weight(std::string(name) + "/kernel", {3, 3, 3, filters, x->C});
            Param* bi = add_param(std::string(name) + "/bias", {filters}, true, INIT_ZEROS);
            BN* bn = add_bn(bn_name, filters, true);
            Act* y = deconv(name, x, k, bi, k333, s, filters, bn, "");
            bn_apply(bn_name, 0, y, bn, nullptr, nullptr, new_view(cat, coff, filters, ""), "");
        };
        up("deconv_pool2", "deconv_pool2_bn", pool2, 2 * b, s111, 0);
        up("deconv_pool3", "deconv_pool3_bn", pool3, 4 * b, s222, 2 * b);
        up("deconv_pool4", "deconv_pool4_bn", pool4, 8 * b, s444, 6 * b);
        Param* kc = conv_weight("conv_concat/kernel", {3, 3, 3, 14 * b, 8 * b});
        Param* bc = add_param("conv_concat/bias", {8 * b}, true, INIT_ZEROS);
        BN* bnc = add_bn("conv_concat_bn", 8 * b, true);
        Act* yc = conv("conv_concat", cat, kc, bc, k333, s111, 8 * b, bnc, "");
        Act* zc = bn_apply("conv_concat_bn", 0, yc, bnc, nullptr, nullptr, nullptr, "conv_concat");
        Param* kr = conv_weight("deconv_revise/kernel", {3, 3, 3, 2 * b, 8 * b});
        Param* br = add_param("deconv_revise/bias", {2 * b}, true, INIT_ZEROS);
        BN* bnr = add_bn("deconv1_revise_bn", 2 * b, true);
        Act* yr = deconv("deconv_revise", zc, kr, br, k333, s222, 2 * b, bnr, "", /*bn_has_dropout=*/true);
        Act* zr = bn_apply("deconv1_revise_bn", 0, yr, bnr, nullptr, nullptr, nullptr, "deconv1_revise", /*dropout=*/true);
        Param* kp = conv_weight("predict_revise/kernel", {3, 3, 3, 1, 2 * b});
        Param* bp = add_param("predict_revise/bias", {1}, true, INIT_ZEROS);
        head(zr, kp, bp, /*with_sigmoid=*/false);
    }

    // ---- self attention, utils/network.py:157-192 (mode 'bn', sub_size 2) --------------------------------------
    struct AttnParams { Param *wf, *bf, *wg, *bg, *wh, *bh, *wo, *bo, *gamma; BN* bn; int ch; std::string name; };
    // p3d_set_attention_mode: 0 = per site (flash where the score matrix of the site has at least attn_flash_min_scores elements),
    // 1 = GEMMs around stored scores everywhere, 2 = flash wherever the kernels exist (ch in 32..256)
    int attn_mode = 0;
    int64_t attn_flash_min_scores = (int64_t)1 << 24;         // 64 MiB of scores per buffer (measured crossover: DESIGN.md)
    std::vector<std::function<void()>> attn_gemm_alloc;
    // variables in the reference's creation order: name/conv3d{,_1,_2}, an unnamed top-level conv3d, an unnamed
    // batch_normalization, then the top-level scalar 'gamma'+name (initialised to 0)
    AttnParams attn_declare(const std::string& name, int ch) {
        AttnParams a;
        a.name = name; a.ch = ch;
        const int ci = std::max(1, ch / 8);
        if (ci % 4) throw P3dError("attention needs channel counts that are multiples of 32 (base multiple of 16)");
        a.wf = conv_weight(name + "/conv3d/kernel", {1, 1, 1, ch, ci});   a.bf = add_param(name + "/conv3d/bias", {ci}, true, INIT_ZEROS);
        a.wg = conv_weight(name + "/conv3d_1/kernel", {1, 1, 1, ch, ci}); a.bg = add_param(name + "/conv3d_1/bias", {ci}, true, INIT_ZEROS);
        a.wh = conv_weight(name + "/conv3d_2/kernel", {1, 1, 1, ch, ch}); a.bh = add_param(name + "/conv3d_2/bias", {ch}, true, INIT_ZEROS);
        const std::string oc = unique("conv3d");
        a.wo = conv_weight(oc + "/kernel", {1, 1, 1, ch, ch}); a.bo = add_param(oc + "/bias", {ch}, true, INIT_ZEROS);
        a.bn = add_bn("", ch, true);
        a.gamma = add_param("gamma" + name, {1}, true, INIT_ZEROS);
        return a;
    }
    // x -> relu(bn(conv(softmax(g f^T) h))) * gamma + x; with subsample the keys f and values h are max-pooled by 2
    // (utils/network.py:178-181; the query pool has size sub_size/2 = 1).  `dropout` folds the tf.layers.dropout
    // that follows the last block (p3d.py:388) into the mixing pass.
    Act* attn_run(const AttnParams& ap, Act* x, bool subsample, const std::string& out_name, bool dropout = false) {
        const int one[3] = {1, 1, 1}, two[3] = {2, 2, 2};
        const int ch = ap.ch, ci = ch / 8, B = x->N;
        if (x->C != ch) throw P3dError("attention declared for another channel count");
        const std::string& nm = ap.name;
        Act* f = conv(nm + "/f", x, ap.wf, ap.bf, one, one, ci, nullptr, "");
        Act* gq = conv(nm + "/g", x, ap.wg, ap.bg, one, one, ci, nullptr, "");
        Act* h = conv(nm + "/h", x, ap.wh, ap.bh, one, one, ch, nullptr, "");
        if (subsample) {
            if ((x->D | x->H | x->W) & 1) throw P3dError("attention sub-sampling needs even extents ('valid' pooling)");
            f = maxpool(nm + "/pool_f", f, two, two, nullptr, "");
            h = maxpool(nm + "/pool_h", h, two, two, nullptr, "");
        }
        const int Ng = gq->D * gq->H * gq->W, Nf = f->D * f->H * f->W, Nfp = (Nf + 3) / 4 * 4;
        const bool pad = Nfp != Nf;
        Act* o = new_act(nm + "/o", B, x->D, x->H, x->W, ch);
        // Two executions of the core (p3d_set_attention_mode): score tiles recomputed on chip (attention_flash.hip: lse + row
        // dots are all it keeps), or three GEMMs per direction around a stored score matrix.  The GEMM path's buffers
        // (2 x B*Ng*Nf floats) are only allocated for sites that may take it.
        const bool can_flash = p3d_flash_attn_ok(ch);
        const bool auto_flash = can_flash && (int64_t)B * Ng * Nfp >= attn_flash_min_scores;
        float* lse = can_flash ? dalloc<float>((int64_t)B * Ng) : nullptr;
        float* dsum = can_flash ? dalloc<float>((int64_t)B * Ng) : nullptr;
        struct GemmBufs { float *sbuf = nullptr, *dsbuf = nullptr, *fpad = nullptr, *hpad = nullptr, *dfpad = nullptr, *dhpad = nullptr; };
        GemmBufs* gb = new GemmBufs();           // lives as long as the handle (ops capture it)
        auto ensure_gemm = [=]() {
            if (gb->sbuf) return;
            gb->sbuf = dalloc<float>((int64_t)B * Ng * Nfp);       // scores, then the attention map beta (kept for backward)
            gb->dsbuf = dalloc<float>((int64_t)B * Ng * Nfp);      // d beta, then d scores
            if (pad) {
                gb->fpad = dalloc<float>((int64_t)B * Nfp * ci); gb->hpad = dalloc<float>((int64_t)B * Nfp * ch);
                gb->dfpad = dalloc<float>((int64_t)B * Nfp * ci); gb->dhpad = dalloc<float>((int64_t)B * Nfp * ch);
            }
        };
        if (!auto_flash) ensure_gemm();
        attn_gemm_alloc.push_back(ensure_gemm);
        bool* ran_flash = new bool(false);       // what the last forward of this site ran (its backward follows)
        auto flash_args = [=]() {
            FlashAttnArgs a;
            memset(&a, 0, sizeof(a));
            a.B = B; a.Ng = Ng; a.Nf = Nf; a.ch = ch;
            a.g = gq->p; a.ldg = gq->ld; a.f = f->p; a.ldf = f->ld; a.h = h->p; a.ldh = h->ld;
            a.o = o->p; a.ldo = o->ld; a.lse = lse;
            a.d_o = o->g; a.lddo = o->ld; a.dsum = dsum;
            a.dg = gq->g; a.lddg = gq->ld; a.df = f->g; a.lddf = f->ld; a.dh = h->g; a.lddh = h->ld;
            return a;
        };
        char* flg = consume(gq); char* flf = consume(f); char* flh = consume(h);
        {
            Op op;
            op.name = nm + "/core"; op.kind = "attention";
            op.flops = 2.0 * B * (double)Ng * Nf * (ci + ch);
            op.bytes = 4.0 * B * ((double)Ng * Nfp * 4 + (double)Ng * (ci + ch) + (double)Nf * (ci + ch));
            op.bflops = 2 * op.flops; op.bbytes = 2 * op.bytes;
            const int gD = gq->D, gH = gq->H, gW = gq->W;
            // one GEMM per clip; if the planner slices K (few rows), the whole output is zeroed once and the launches add
            auto gemm_each = [=](const Ctx& c, float* out, int ldo, int Nc, std::function<IgemmArgs(int)> mk) {
                IgemmArgs t = mk(0);
                const bool split = p3d_igemm2_plan(t, 1).splits > 1;
                if (split) zero_strided(c, out, ldo, (int64_t)B * Ng, Nc);
                for (int b = 0; b < B; ++b) launch_igemm(c, mk(b), split ? 1 : 0);
            };
            const double pair_flops = 2.0 * B * (double)Ng * Nf * (ci + ch);
            const double operand_bytes = 4.0 * B * ((double)Ng * (ci + ch) + (double)Nf * (ci + ch));
            op.fwd = [=](const Ctx& c) {
                *ran_flash = can_flash && (attn_mode == 2 || (attn_mode == 0 && auto_flash));
                if (*ran_flash) {
                    launch(c, "flash_fwd_kernel", pair_flops, operand_bytes, [&]() { return p3d_flash_attn_fwd(flash_args(), c.s); });
                    return;
                }
                if (!gb->sbuf) throw P3dError("attention GEMM path without its score buffers (p3d_set_attention_mode allocates them)");
                float* const sbuf = gb->sbuf; float* const fpad = gb->fpad; float* const hpad = gb->hpad;
                const float* F = f->p; const float* H = h->p;
                if (pad) {
                    launch(c, "pad_rows_kernel", 0, 8.0 * B * Nfp * ci, [&]() { return p3d_pad_rows(f->p, fpad, B, Nf, Nfp, ci, c.s); });
                    launch(c, "pad_rows_kernel", 0, 8.0 * B * Nfp * ch, [&]() { return p3d_pad_rows(h->p, hpad, B, Nf, Nfp, ch, c.s); });
                    F = fpad; H = hpad;
                }
                gemm_each(c, sbuf, Nfp, Nfp, [=](int b) {
                    return gemm_rows(gD, gH, gW, gq->p + (int64_t)b * Ng * gq->ld, gq->ld, ci, F + (int64_t)b * Nfp * ci, 1,
                                     sbuf + (int64_t)b * Ng * Nfp, Nfp, Nfp);
                });
                launch(c, "softmax_fwd_kernel", 0, 8.0 * B * Ng * Nfp, [&]() { return p3d_softmax_rows(sbuf, (long long)B * Ng, Nf, Nfp, c.s); });
                gemm_each(c, o->p, o->ld, ch, [=](int b) {
                    return gemm_rows(gD, gH, gW, sbuf + (int64_t)b * Ng * Nfp, Nfp, Nfp, H + (int64_t)b * Nfp * ch, 0,
                                     o->p + (int64_t)b * Ng * o->ld, o->ld, ch);
                });
            };
            op.bwd = [=](const Ctx& c) {
                if (*flg || *flf || *flh) throw P3dError("attention operands have one consumer each");
                if (*ran_flash) {
                    if (c.dry) return;
                    const FlashAttnArgs a = flash_args();
                    // one entry point, three launches (row dots; dg per query tile; df, dh per key tile)
                    launch(c, "flash_bwd(rowdot + q + k kernels)", 3.2 * pair_flops, 3 * operand_bytes, [&]() { return p3d_flash_attn_bwd(a, c.s); });
                    return;
                }
                float* const sbuf = gb->sbuf; float* const dsbuf = gb->dsbuf;
                float* const fpad = gb->fpad; float* const hpad = gb->hpad; float* const dfpad = gb->dfpad; float* const dhpad = gb->dhpad;
                const float* F = pad ? fpad : f->p; const float* H = pad ? hpad : h->p;
                float* dF = pad ? dfpad : f->g; float* dH = pad ? dhpad : h->g;
                gemm_each(c, dsbuf, Nfp, Nfp, [=](int b) {          // d beta = d o * h^T
                    return gemm_rows(gD, gH, gW, o->g + (int64_t)b * Ng * o->ld, o->ld, ch, H + (int64_t)b * Nfp * ch, 1,
                                     dsbuf + (int64_t)b * Ng * Nfp, Nfp, Nfp);
                });
                zero_strided(c, dH, ch, (int64_t)B * Nfp, ch);
                for (int b = 0; b < B; ++b)                        // d h = beta^T * d o
                    launch_wgrad(c, gemm_tn(gD, gH, gW, sbuf + (int64_t)b * Ng * Nfp, Nfp, Nfp, o->g + (int64_t)b * Ng * o->ld, o->ld, ch,
                                            dH + (int64_t)b * Nfp * ch));
                launch(c, "softmax_bwd_kernel", 0, 12.0 * B * Ng * Nfp, [&]() { return p3d_softmax_rows_bwd(sbuf, dsbuf, (long long)B * Ng, Nf, Nfp, c.s); });
                gemm_each(c, gq->g, gq->ld, ci, [=](int b) {        // d g = d s * f
                    return gemm_rows(gD, gH, gW, dsbuf + (int64_t)b * Ng * Nfp, Nfp, Nfp, F + (int64_t)b * Nfp * ci, 0,
                                     gq->g + (int64_t)b * Ng * gq->ld, gq->ld, ci);
                });
                zero_strided(c, dF, ci, (int64_t)B * Nfp, ci);
                for (int b = 0; b < B; ++b)                        // d f = d s^T * g
                    launch_wgrad(c, gemm_tn(gD, gH, gW, dsbuf + (int64_t)b * Ng * Nfp, Nfp, Nfp, gq->p + (int64_t)b * Ng * gq->ld, gq->ld, ci,
                                            dF + (int64_t)b * Nfp * ci));
                if (pad) {
                    launch(c, "unpad_rows_kernel", 0, 8.0 * B * Nf * ci, [&]() { return p3d_unpad_rows(dfpad, f->g, B, Nf, Nfp, ci, c.s); });
                    launch(c, "unpad_rows_kernel", 0, 8.0 * B * Nf * ch, [&]() { return p3d_unpad_rows(dhpad, h->g, B, Nf, Nfp, ch, c.s); });
                }
            };
            ops.push_back(op);
        }
        Act* y = conv(nm + "/out", o, ap.wo, ap.bo, one, one, ch, ap.bn, "");
        Act* r = bn_apply(nm + "/out_bn", 0, y, ap.bn, nullptr, nullptr, nullptr, "");
        Act* z = new_act(out_name, B, x->D, x->H, x->W, ch);
        char* flr = consume(r); char* flx = consume(x);
        {
            Op op;
            op.name = nm + "/mix"; op.kind = "attention_mix";
            op.bytes = 4.0 * 3 * x->rows() * ch; op.bbytes = 4.0 * 5 * x->rows() * ch;
            op.owns = {ap.gamma};
            Param* gamma = ap.gamma;
            auto mk = [=](const Ctx& c) {
                AttnMixArgs a;
                memset(&a, 0, sizeof(a));
                a.M = x->rows(); a.C = ch; a.r = r->p; a.ldr = r->ld; a.x = x->p; a.ldx = x->ld; a.gamma = gamma->p;
                a.z = z->p; a.ldz = z->ld; a.dz = z->g; a.dr = r->g; a.dx = x->g; a.accx = *flx; a.dgamma = gamma->g;
                if (dropout && c.training && c.drop > 0.f) { a.drop_rate = c.drop; a.drop_scale = 1.f / (1.f - c.drop); a.seed = c.seed; a.seed_dev = c.seed_dev; }
                return a;
            };
            const double fb = op.bytes, bb = op.bbytes;
            op.fwd = [=](const Ctx& c) { launch(c, "mix_fwd_kernel", 0, fb, [&]() { return p3d_attn_mix_fwd(mk(c), c.s); }); };
            op.bwd = [=](const Ctx& c) {
                if (*flr) throw P3dError("attention branch has one consumer");
                launch(c, "mix_bwd_kernel", 0, bb, [&]() { return p3d_attn_mix_bwd(mk(c), c.s); });
            };
            ops.push_back(op);
        }
        return z;
    }

    // p3d_unetplusplus_nonsa (p3d.py:401-459): the nested UNet++ head without the attention blocks.  Every layer
    // is utils/network.py:100-110: named tf.layers.conv3d / conv3d_transpose + an UNNAMED batch_normalization
    // (it follows `training` and continues the backbone's counter in the reference's call order) + ReLU.
    // Variables are therefore created in the reference's order, but the ops run in an order in which each
    // concat buffer's whole consumer (the x_i_j conv) comes after every consumer of one of its slices -- see
    // consume().  Concats are zero-copy: producers write channel slices of the buffers below.
    // with sa = true: p3d_unetplusplus_ds (p3d.py:340-397), the same head with attention() on x_4_0, x_3_1, x_2_2
    // and (keys / values pooled by 2, followed by the dropout) x_1_3
    void build_unetpp(bool sa) {
        const int B = cfg.batch, T = cfg.frames, H = cfg.height, W = cfg.width, b = cfg.base;
        const int s111[3] = {1, 1, 1}, s222[3] = {2, 2, 2}, k211[3] = {2, 1, 1}, s211[3] = {2, 1, 1};
        Act* cat31 = new_act("cat_x_3_1", B, T / 8, H / 8, W / 8, 16 * b);    // [x_3_0 | upx_4_0]
        Act* cat21 = new_act("cat_x_2_1", B, T / 4, H / 4, W / 4, 8 * b);     // [x_2_0 | upx_3_0]
        Act* cat22 = new_act("cat_x_2_2", B, T / 4, H / 4, W / 4, 8 * b);     // [x_2_1 | upx_3_1]
        Act* cat11 = new_act("cat_x_1_1", B, T / 2, H / 2, W / 2, 3 * b);     // [x_1_0 | upx_2_0]
        Act* cat12 = new_act("cat_x_1_2", B, T / 2, H / 2, W / 2, 4 * b);     // [x_1_1 | upx_2_1]
        Act* cat13 = new_act("cat_x_1_3", B, T / 2, H / 2, W / 2, 4 * b);     // [x_1_2 | upx_2_2]
        Act *x_2_0, *x_3_0, *x_4_0;
        build_encoder(new_view(cat21, 0, 4 * b, "pool2"), new_view(cat31, 0, 8 * b, "pool3"), x_2_0, x_3_0, x_4_0);
        Act* x_1_0 = maxpool("x_1_0", stem_out, k211, s211, new_view(cat11, 0, b, ""), "x_1_0");
        (void)x_1_0;
        struct Layer { Param *k, *bias; BN* bn; int filters; int kk[3]; bool up; };
        std::map<std::string, Layer> L;
        auto declare = [&](const char* name, bool up, int cin, int filters, int kd) {
            Layer l;
            l.up = up; l.filters = filters; l.kk[0] = kd; l.kk[1] = 3; l.kk[2] = 3;
            l.k = up ? conv_weight(std::string(name) + "/kernel", {kd, 3, 3, filters, cin})
                     : conv_weight(std::string(name) + "/kernel", {kd, 3, 3, cin, filters});
            l.bias = add_param(std::string(name) + "/bias", {filters}, true, INIT_ZEROS);
            l.bn = add_bn("", filters, true);
            L[name] = l;
        };
        // reference creation order (p3d.py:371-387 / 435-451)
        AttnParams sa40, sa31, sa22, sa13;
        if (sa) sa40 = attn_declare("x_4_0_sa", 16 * b);
        declare("upx_4_0", true, 16 * b, 8 * b, 1);
        declare("x_3_1", false, 16 * b, 8 * b, 2);
        if (sa) sa31 = attn_declare("x_3_1_sa", 8 * b);
        declare("upx_3_0", true, 8 * b, 4 * b, 2);
        declare("x_2_1", false, 8 * b, 4 * b, 3);
        declare("upx_3_1", true, 8 * b, 4 * b, 2);
        declare("x_2_2", false, 8 * b, 4 * b, 3);
        if (sa) sa22 = attn_declare("x_2_2_sa", 4 * b);
        declare("upx_2_0", true, 4 * b, 2 * b, 3);
        declare("x_1_1", false, 3 * b, 2 * b, 3);
        declare("upx_2_1", true, 4 * b, 2 * b, 3);
        declare("x_1_2", false, 4 * b, 2 * b, 3);
        declare("upx_2_2", true, 4 * b, 2 * b, 3);
        declare("x_1_3", false, 4 * b, 2 * b, 3);
        if (sa) sa13 = attn_declare("x_1_3_sa", 2 * b);
        auto run = [&](const char* name, Act* x, Act* out, bool dropout = false) -> Act* {
            const Layer& l = L.at(name);
            Act* y = l.up ? deconv(name, x, l.k, l.bias, l.kk, s222, l.filters, l.bn, "", dropout)
                          : conv(name, x, l.k, l.bias, l.kk, s111, l.filters, l.bn, "", false, dropout);
            return bn_apply(std::string(name) + "_bn", 0, y, l.bn, nullptr, nullptr, out, name, dropout);
        };
        if (sa) x_4_0 = attn_run(sa40, x_4_0, false, "x_4_0_sa");
        run("upx_4_0", x_4_0, new_view(cat31, 8 * b, 8 * b, ""));
        run("upx_3_0", x_3_0, new_view(cat21, 4 * b, 4 * b, ""));
        run("upx_2_0", x_2_0, new_view(cat11, b, 2 * b, ""));
        Act* x_3_1 = run("x_3_1", cat31, nullptr);
        if (sa) x_3_1 = attn_run(sa31, x_3_1, false, "x_3_1_sa");
        Act* x_2_1 = run("x_2_1", cat21, new_view(cat22, 0, 4 * b, ""));
        run("x_1_1", cat11, new_view(cat12, 0, 2 * b, ""));
        run("upx_3_1", x_3_1, new_view(cat22, 4 * b, 4 * b, ""));
        run("upx_2_1", x_2_1, new_view(cat12, 2 * b, 2 * b, ""));
        Act* x_2_2 = run("x_2_2", cat22, nullptr);
        if (sa) x_2_2 = attn_run(sa22, x_2_2, false, "x_2_2_sa");
        run("x_1_2", cat12, new_view(cat13, 0, 2 * b, ""));
        run("upx_2_2", x_2_2, new_view(cat13, 2 * b, 2 * b, ""));
        Act* x_1_3 = run("x_1_3", cat13, nullptr, /*dropout=*/!sa);
        if (sa) x_1_3 = attn_run(sa13, x_1_3, true, "x_1_3_sa", /*dropout=*/true);
        Param* kh = conv_weight("x_0_1/kernel", {3, 3, 3, 1, 2 * b});
        Param* bh = add_param("x_0_1/bias", {1}, true, INIT_ZEROS);
        head(x_1_3, kh, bh);
    }

    // results = sigmoid(conv3d_transpose(x, 1, 3, 2)) (p3d.py:217-219) + Smooth-L1 (train.py:156-159)
    bool head_sigmoid = true;
    // transpose = false: the stride-1 tf.layers.conv3d(x, 1, 3, 1, 'same') of gn/p3d_gn.py:537 instead
    void head(Act* x, Param* k, Param* bias, bool with_sigmoid = true, bool transpose = true) {
        head_sigmoid = with_sigmoid;
        const int up = transpose ? 2 : 1;
        logits = new_act("logits", x->N, up * x->D, up * x->H, up * x->W, 1, false);
        pred = new_act("pred", x->N, up * x->D, up * x->H, up * x->W, 1, false);
        d_dlogits = dalloc<float>(pred->rows());
        d_y = dalloc<float>(pred->rows());
        d_loss = dalloc<double>(1);
        char* xflag = consume(x);
        Op op;
        op.name = "results"; op.kind = transpose ? "head_deconv" : "head_conv";
        op.flops = 2.0 * x->rows() * 27 * x->C;
        op.bytes = 4.0 * (x->rows() * (double)x->C + 2.0 * pred->rows());
        op.bflops = 2 * op.flops; op.bbytes = 4.0 * (3.0 * x->rows() * (double)x->C + 2.0 * pred->rows());
        op.owns = {k, bias};
        auto mk = [=]() {
            HeadArgs a;
            memset(&a, 0, sizeof(a));
            a.x = x->p; a.N = x->N; a.D = x->D; a.H = x->H; a.W = x->W; a.C = x->C;
            a.k = k->p; a.bias = bias->p; a.logits = logits->p; a.pred = pred->p; a.sigmoid = with_sigmoid;
            a.dlogits = d_dlogits; a.dx = x->g; a.dk = k->g; a.dbias = bias->g;
            return a;
        };
        const double hf = op.flops, hb = op.bytes;
        hipEvent_t head_fork = new_fork_event();
        op.fwd = [=](const Ctx& c) {
            if (transpose) launch(c, "head_fwd_kernel", hf, hb, [&]() { return p3d_head_fwd(mk(), c.s); });
            else launch(c, "headc_fwd_kernel", hf, hb, [&]() { return p3d_headc_fwd(mk(), c.s); });
        };
        op.bwd = [=](const Ctx& c) {
            if (*xflag) throw P3dError("head input gradient must be the first writer");
            // the filter gradient is a weight gradient like any other: side stream, off the critical path
            on_side_stream(c, head_fork, [=](const Ctx& sc) {
                if (transpose) launch(sc, "head_bwd_filter_kernel", hf, hb, [&]() { return p3d_head_bwd_filter(mk(), sc.s); });
                else launch(sc, "headc_bwd_filter_kernel", hf, hb, [&]() { return p3d_headc_bwd_filter(mk(), sc.s); });
            });
            if (transpose) launch(c, "head_bwd_input_kernel", hf, hb, [&]() { return p3d_head_bwd_input(mk(), c.s); });
            else launch(c, "headc_bwd_input_kernel", hf, hb, [&]() { return p3d_headc_bwd_input(mk(), c.s); });
        };
        ops.push_back(op);
    }

    void finalize_build() {
        flat_p = dalloc<float>(n_train); flat_g = dalloc<float>(n_train);
        flat_m = dalloc<float>(n_train); flat_v = dalloc<float>(n_train);
        flat_state = dalloc<float>(n_state);
        HIPCHECK(hipMemset(flat_p, 0, (size_t)n_train * 4)); HIPCHECK(hipMemset(flat_g, 0, (size_t)n_train * 4));
        HIPCHECK(hipMemset(flat_m, 0, (size_t)n_train * 4)); HIPCHECK(hipMemset(flat_v, 0, (size_t)n_train * 4));
        HIPCHECK(hipMemset(flat_state, 0, (size_t)n_state * 4));
        for (Param* p : porder) {
            if (p->trainable) { p->p = flat_p + p->off; p->g = flat_g + p->off; }
            else p->p = flat_state + p->off;
        }
        stats_arena = dalloc<double>(stats_count);
        statpart_arena = dalloc<float>(statpart_count);
        d_seed = dalloc<unsigned long long>(1);
        d_lr = dalloc<float>(1);
        red_arena = dalloc<double>(red_count);
        bnbuf = dalloc<float>(bnbuf_count);
        for (auto& f : late_bind) f();
        late_bind.clear();
        index_gradient_owners();
        for (int i = 0; i < (int)ops.size(); ++i)              // last op of the encoder's last bottleneck (ops are named blockN/...)
            if (ops[i].name.compare(0, 5, "block") == 0) defer_release_op = i;
        if (defer_release_op == (int)ops.size() - 1) defer_release_op = -1;
        // the encoder's last stage leaves CUs idle only while its tensors are small (784 rows at 8 clips of 16x112x112); the
        // budget is the decoder filter-gradient work of the unet at that size (124 GFLOP), measured to be absorbed
        defer_budget = 0;
        if (!blocks.empty()) {
            const Act* last = blocks.rbegin()->second.out;
            if (last && last->rows() <= 2048) defer_budget = 130e9;
            if (const char* e = p3d_tune_env("P3D_TUNE_DEFER_GFLOP")) defer_budget = atof(e) * 1e9;      // A/B runs
        }
        tune_plans();
        plan_zero_arenas();
    }

    // Bucketed all-reduce needs to know, after the backward of op i, the lowest flat offset above which every
    // gradient is final.  Variables are laid out in TF creation order, which need not be op order (the unet++
    // head creates its layers in the reference's order but runs them in a concat-safe order), so index the
    // owners: own_sorted = (offset, op index) ascending by offset, own_sufmin[p] = min op index over [p, end).
    std::vector<std::pair<int64_t, int>> own_sorted;
    std::vector<int> own_sufmin;
    void index_gradient_owners() {
        std::map<const Param*, int> owner;
        for (size_t i = 0; i < ops.size(); ++i)
            for (Param* p : ops[i].owns) {
                if (!p || !p->trainable) continue;
                if (owner.count(p)) throw P3dError("variable " + p->name + " has two gradient producers");
                owner[p] = (int)i;
            }
        for (Param* p : porder)
            if (p->trainable && !owner.count(p)) throw P3dError("variable " + p->name + " has no gradient producer");
        own_sorted.clear();
        for (auto& kv : owner) own_sorted.push_back({kv.first->off, kv.second});
        std::sort(own_sorted.begin(), own_sorted.end());
        own_sufmin.assign(own_sorted.size(), 0);
        int m = (int)ops.size();
        for (size_t p = own_sorted.size(); p-- > 0;) { m = std::min(m, own_sorted[p].second); own_sufmin[p] = m; }
        // brute-force check of the walk run_backward does: after op i, nothing at or above `lo` may belong to an
        // op that has not run its backward yet (a premature all-reduce would silently drop gradient terms)
        size_t pos = own_sorted.size();
        for (int i = (int)ops.size() - 1; i >= 0; --i) {
            while (pos > 0 && own_sufmin[pos - 1] >= i) --pos;
            const int64_t lo = pos < own_sorted.size() ? own_sorted[pos].first : n_train;
            for (auto& kv : owner)
                if (kv.first->off >= lo && kv.second < i)
                    throw P3dError("gradient bucket order broken at op " + ops[i].name + " / variable " + kv.first->name);
        }
        if (pos != 0) throw P3dError("gradient bucket walk does not reach offset 0");
        // split point of the two-part optimiser step (run_backward): the lowest offset above which no variable belongs to op 0;
        // usable when it is 16-byte aligned and op 0 (the stem conv) really owns something below it
        adam_split = 0;
        size_t q = own_sorted.size();
        while (q > 0 && own_sufmin[q - 1] >= 1) --q;
        if (q > 0 && q < own_sorted.size() && (own_sorted[q].first & 3) == 0) adam_split = own_sorted[q].first;
    }

    // One forward + backward over whatever the buffers hold: sizes the per-stream scratch of the K-sliced launches
    // and sets the kernels' function attributes, so that nothing allocates later (a captured step graph must not).
    // Parameters and moving statistics are not touched.
    void tune_plans() {
        Ctx c; c.training = true; c.s = stream;
        const bool want = fuse_bn;
        const bool want_bwd = fuse_bn_bwd;
        for (int mode = 0; mode < 3; ++mode) {      // every launch list: BatchNorm fusion can be switched per handle later
            fuse_bn = mode >= 1; fuse_bn_bwd = mode == 2;
            run_forward(c);
            run_loss(c);
            run_backward(c, false);
        }
        fuse_bn = want; fuse_bn_bwd = want_bwd;
        HIPCHECK(hipStreamSynchronize(c.s));
        HIPCHECK(hipMemsetAsync(flat_g, 0, (size_t)n_train * sizeof(float), c.s));
        HIPCHECK(hipStreamSynchronize(c.s));
    }

    // Dry-run forward and backward once: every dense buffer an op would zero-fill before adding into it
    // (split-K outputs, atomically scattered gradients) is moved into one contiguous arena per phase, so a
    // step issues two memsets instead of a few hundred.
    void plan_zero_arenas() {
        for (int phase = 0; phase < 2; ++phase) {
            std::vector<std::pair<float*, size_t>> reqs;
            Ctx c; c.training = true; c.s = stream; c.dry = &reqs;
            if (phase == 0) { for (auto& op : ops) op.fwd(c); }
            else { for (int i = (int)ops.size() - 1; i >= 0; --i) ops[i].bwd(c); }
            std::vector<std::pair<Act*, bool>> movers;     // (act, is_grad)
            size_t total = 0;
            for (auto& rq : reqs)
                for (auto& a : acts) {
                    if (a.parent || !a.views.empty()) continue;
                    const size_t bytes = (size_t)a.rows() * a.C * sizeof(float);
                    if (bytes != rq.second) continue;
                    const bool is_g = (a.g == rq.first), is_p = (a.p == rq.first);
                    if (!is_g && !is_p) continue;
                    movers.push_back({&a, is_g});
                    total += (bytes + 255) / 256 * 256;
                    break;
                }
            if (!total) continue;
            char* base = (char*)dalloc<char>((int64_t)total);
            size_t off = 0;
            for (auto& mv : movers) {
                float*& ptr = mv.second ? mv.first->g : mv.first->p;
                ptr = (float*)(base + off);                 // the old allocation stays owned by `allocs`
                off += ((size_t)mv.first->rows() * mv.first->C * sizeof(float) + 255) / 256 * 256;
            }
            if (phase == 0) { zf = base; zf_bytes = total; } else { zb = base; zb_bytes = total; }
        }
    }

    // ---- execution -------------------------------------------------------------------------------
    // P3D_DEBUG_SYNC=1: synchronise and log after every op (fault isolation, not for timing)
    void debug_sync(const char* dir, const Op& op, const Ctx& c) {
        if (!runtime_env().debug_sync) return;
        fprintf(stderr, "[p3d] %s %s (%s) ...", dir, op.name.c_str(), op.kind.c_str());
        fflush(stderr);
        HIPCHECK(hipStreamSynchronize(c.s));
        fprintf(stderr, " ok\n");
        fflush(stderr);
    }
    void run_forward(const Ctx& c) {
        sib_pending.clear();          // (a pass that threw between the two siblings of an ST_B pair must not hand its launch to this one)
        if (stats_count) HIPCHECK(hipMemsetAsync(stats_arena, 0, (size_t)stats_count * sizeof(double), c.s));
        if (zf_bytes) HIPCHECK(hipMemsetAsync(zf, 0, zf_bytes, c.s));
        Ctx cz = c; cz.z0 = zf; cz.z1 = zf + zf_bytes;
        cz.fuse = fuse_bn && !c.per_sample && !c.dry;
        last_forward_fused = cz.fuse;
        const bool no_side_f = runtime_env().no_side_stream;
        cz.side = (c.prof || no_side_f || c.dry) ? nullptr : side_stream;      // ST_B sibling convs overlap
        const Ctx& c2 = cz;
        for (auto& op : ops) {
            if (c.prof) c.prof->cur_op = op.name;
            op.fwd(c2);
            debug_sync("fwd", op, c);
        }
    }
    void run_loss(const Ctx& c) {
        HIPCHECK(hipMemsetAsync(d_loss, 0, sizeof(double), c.s));
        launch(c, "smooth_l1_kernel", 0, 12.0 * pred->rows(), [&]() { return p3d_smooth_l1(pred->p, d_y, pred->rows(), d_loss, d_dlogits, head_sigmoid ? 1 : 0, c.s); });
    }
    // with_adam: the optimiser step is part of the call and split in two -- every variable but the first op's is updated while
    // that op's filter gradient (the stem's: the last launch of the pass, alone on the side stream) is still running, the
    // first op's own variables after it.  Returns whether Adam ran (false: the caller launches run_adam).
    // Gradient buffer (248 MB), gradient arena of the activations and the double-precision reduction arena start a backward pass
    // at zero.  A train step knows that a backward pass follows its forward pass: it zeroes them on the side stream while the
    // forward runs (the previous step's optimiser has read the gradients: the side stream is joined before it) instead of
    // ~0.1 ms of fills at the head of the backward on the main stream.
    bool zeroed_early = false;
    hipEvent_t ev_zeroed = nullptr, ev_zero_fork = nullptr;
    void zero_backward_arenas(hipStream_t st, bool with_zb) {
        // (zb holds whatever buffers the backward ops zero-fill before adding into them; it is planned from a dry run and may
        //  name a buffer the forward pass also touches, so it is never zeroed early)
        if (with_zb && zb_bytes) HIPCHECK(hipMemsetAsync(zb, 0, zb_bytes, st));
        HIPCHECK(hipMemsetAsync(flat_g, 0, (size_t)n_train * sizeof(float), st));
        if (red_count) HIPCHECK(hipMemsetAsync(red_arena, 0, (size_t)red_count * sizeof(double), st));
    }
    void zero_early(const Ctx& c) {      // call right before run_forward of a train step (not while capturing, not when profiling)
        if (runtime_env().no_side_stream || !side_stream || c.prof || c.dry) return;
        if (!ev_zeroed) HIPCHECK(hipEventCreateWithFlags(&ev_zeroed, local_event_flags()));
        // after everything the main stream has queued so far (the previous step's optimiser and whoever read the gradients)
        if (!ev_zero_fork) HIPCHECK(hipEventCreateWithFlags(&ev_zero_fork, local_event_flags()));
        HIPCHECK(hipEventRecord(ev_zero_fork, c.s));
        HIPCHECK(hipStreamWaitEvent(side_stream, ev_zero_fork, 0));
        zero_backward_arenas(side_stream, false);
        HIPCHECK(hipEventRecord(ev_zeroed, side_stream));
        zeroed_early = true;
    }
    bool run_backward(const Ctx& c0, bool allreduce, bool with_adam = false) {
        Ctx c = c0; c.z0 = zb; c.z1 = zb + zb_bytes;
        c.fuse = last_forward_fused && !c.dry;       // the backward follows the forward that produced the activations
        c.fuse_bwd = c.fuse && fuse_bn_bwd;
        bool adam_done = false;
        const bool no_side = runtime_env().no_side_stream;
        c.side = (c.prof || no_side) ? nullptr : side_stream;      // per-launch profiling keeps one stream
        if (zeroed_early) {          // the train step zeroed the backward's arenas on the side stream, beside its forward pass
            zeroed_early = false;
            HIPCHECK(hipStreamWaitEvent(c.s, ev_zeroed, 0));
            if (zb_bytes) HIPCHECK(hipMemsetAsync(zb, 0, zb_bytes, c.s));
        } else {
            zero_backward_arenas(c.s, true);
        }
        int64_t hi = n_train;                        // grads in [hi, n_train) are already handed to the comm stream
        size_t own_pos = own_sorted.size();
        wq.clear(); wq_flushes = 0;
        // The decoder's backward saturates the chip (deconv input gradients at 80+ TFLOP/s) while the encoder's is a chain of
        // small launches that leaves most CUs idle.  The decoder's side-stream jobs (filter and bias gradients) are therefore
        // parked and released when the walk reaches the encoder, where they fill idle CUs instead of halving the rate of the
        // main stream's big kernels.  Grouping and summation order do not change, only the launch time.
        // Parking is bounded by what that phase can absorb (defer_budget, finalize_build): a head that is many times the
        // encoder (unet++, the GN nets) parks its first jobs only, and nothing is parked when the encoder's own launches fill
        // the chip (32x224x224 clips).
        std::vector<std::pair<hipEvent_t, std::function<void(const Ctx&)>>> parked;
        bool early_adam = false, early_comm = false;
        static const bool no_defer = p3d_tune_env("P3D_DEFER_SIDE") && atoi(p3d_tune_env("P3D_DEFER_SIDE")) == 0;
        const bool defer_on = c.side && !no_defer && defer_release_op > 0 && defer_budget > 0;
        parked_flops = 0;
        auto release_parked = [&]() {
            c.defer = nullptr;
            for (auto& job : parked) on_side_stream(c, job.first, job.second);
            parked.clear();
        };
        for (int i = (int)ops.size() - 1; i >= 0; --i) {
            if (c.prof) c.prof->cur_op = ops[i].name;
            c.bwd_op = ops[i].name.c_str();
            if (i == 1) flush_wgrads(c);      // what is still queued runs beside the stem's normalisation backward, not after it
            if (i == 0 && adam_split > 0 && adam_split < n_train && c.side && !c.dry) {
                // every gradient at offsets >= adam_split is final once the side stream has drained what is queued so far:
                // hand that range over now, so that neither its all-reduce nor its Adam waits for the first op's filter gradient
                flush_wgrads(c);
                if (c.defer || !parked.empty()) release_parked();
                const bool reduce = allreduce && (comm || bucket_hook);
                if (hi > adam_split) {
                    if (reduce) reduce_range(adam_split, hi, c, 1);
                    hi = adam_split;
                }
                static const bool no_split = p3d_tune_env("P3D_SPLIT_ADAM") && atoi(p3d_tune_env("P3D_SPLIT_ADAM")) == 0;     // A/B runs
                if (with_adam && !no_split) {
                    // what the first Adam part has to wait for is marked NOW, before the first op's filter gradient goes to the
                    // side stream; the part itself is enqueued after that op's backward (below), so that the two overlap --
                    // enqueued here, the filter gradient's fork event would sit behind Adam on the main stream and the tail of
                    // the step would be Adam, then the filter gradient, then the second Adam part, one after the other
                    HIPCHECK(hipEventRecord(ev_side_early, c.side));
                    early_comm = reduce && comm && !bucket_hook;
                    if (early_comm) HIPCHECK(hipEventRecord(ev_comm_early, comm_stream));
                    early_adam = true;
                }
            }
            if (defer_on) {
                if (i > defer_release_op) c.defer = parked_flops < defer_budget ? &parked : nullptr;
                else if (c.defer || !parked.empty()) release_parked();
            }
            ops[i].bwd(c);
            debug_sync("bwd", ops[i], c);
            if (early_adam) {
                early_adam = false;
                HIPCHECK(hipStreamWaitEvent(c.s, ev_side_early, 0));
                if (early_comm) HIPCHECK(hipStreamWaitEvent(c.s, ev_comm_early, 0));
                adam_begin(c);
                adam_range(c, adam_split, n_train);
                adam_done = true;
            }
            {
                // gradients at flat offsets >= lo belong to ops i.. only, so they are final now.  The walk (and the flush
                // of queued filter gradients at every bucket boundary) runs with or without a communicator, so that the
                // grouping of filter gradients -- hence every bit of the result -- does not depend on the world size or on
                // whether the call is a train step or the parity hook p3d_backward.
                while (own_pos > 0 && own_sufmin[own_pos - 1] >= i) --own_pos;
                const int64_t lo = own_pos < own_sorted.size() ? own_sorted[own_pos].first : n_train;
                if ((hi > lo && hi - lo >= bucket_floats) || i == 0) {
                    const int64_t start = (i == 0) ? 0 : lo;
                    flush_wgrads(c);                 // the bucket's queued filter gradients must be on the side stream first
                    if (i == 0 && (c.defer || !parked.empty())) release_parked();
                    // while jobs are parked their gradients are not on the side stream yet: the range stays with the walk
                    // and is handed over at the first boundary after the release
                    if (parked.empty()) {
                        if (hi > start && allreduce && (comm || bucket_hook)) reduce_range(start, hi, c, i);
                        hi = start;
                    }
                }
            }
        }
        flush_wgrads(c);
        if (c.defer || !parked.empty()) release_parked();
        static const bool tune_tail = p3d_tune_env("P3D_TUNE_TAIL") != nullptr;   // diagnostic: how long the side stream outlasts the main one
        static hipEvent_t tail_main = nullptr, tail_side = nullptr;
        if (tune_tail && c.side && !c.dry) {
            if (!tail_main) { HIPCHECK(hipEventCreate(&tail_main)); HIPCHECK(hipEventCreate(&tail_side)); }
            HIPCHECK(hipEventRecord(tail_main, c.s));
            HIPCHECK(hipEventRecord(tail_side, c.side));
        }
        if (c.side) {       // weight gradients must be complete before the optimiser (and the next step)
            HIPCHECK(hipEventRecord(ev_side_done, c.side));
            HIPCHECK(hipStreamWaitEvent(c.s, ev_side_done, 0));
        }
        if (tune_tail && c.side && !c.dry) {
            HIPCHECK(hipEventSynchronize(tail_main)); HIPCHECK(hipEventSynchronize(tail_side));
            float ms = 0.f;
            const hipError_t e = hipEventElapsedTime(&ms, tail_main, tail_side);
            fprintf(stderr, "[p3d tune] side stream ends %.3f ms after the main stream's backward%s (adam_split %lld)\n", e == hipSuccess ? ms : -1.f,
                    adam_done ? " + first Adam part" : "", (long long)adam_split);
        }
        if (allreduce && (comm || bucket_hook)) {
            if (hi > 0) reduce_range(0, hi, c, 0);
            if (comm && !bucket_hook) {
                HIPCHECK(hipEventRecord(ev_comm_done, comm_stream));
                HIPCHECK(hipStreamWaitEvent(c.s, ev_comm_done, 0));
            }
        }
        if (adam_done) adam_range(c, 0, adam_split);
        return adam_done;
    }
    // audit hook (p3d_debug_bucket_audit): called in place of the collective with the range and the op whose backward
    // had just run when the bucket was handed over
    std::function<void(int64_t, int64_t, int)> bucket_hook;
    void reduce_range(int64_t lo, int64_t hi, const Ctx& c, int after_op) {
        if (bucket_hook) { bucket_hook(lo, hi, after_op); return; }
        HIPCHECK(hipEventRecord(ev_bucket, c.s));
        HIPCHECK(hipStreamWaitEvent(comm_stream, ev_bucket, 0));
        if (c.side) {       // the bucket's weight gradients were queued on the side stream
            HIPCHECK(hipEventRecord(ev_side_bucket, c.side));
            HIPCHECK(hipStreamWaitEvent(comm_stream, ev_side_bucket, 0));
        }
        NCCLCHECK(ncclAllReduce(flat_g + lo, flat_g + lo, (size_t)(hi - lo), ncclFloat, ncclSum, comm, comm_stream));
    }
    // tf.train.AdamOptimizer's bias-corrected step size lr * sqrt(1 - b2^t) / (1 - b1^t) for step t (train.py:168)
    float adam_lr_t(int64_t t_step) const {
        const double t = (double)t_step;
        return (float)(lr * std::sqrt(1.0 - std::pow((double)b2, t)) / (1.0 - std::pow((double)b1, t)));
    }
    float cur_lr_t = 0.f;
    int64_t adam_split = 0;              // flat offset below which only the first op's variables live (0: no split)
    hipEvent_t ev_side_early = nullptr, ev_comm_early = nullptr;
    void adam_begin(const Ctx& c) {      // c.lr_dev set: the step size comes from device memory (graph replay), `step` is the caller's
        cur_lr_t = c.lr_dev ? 0.f : adam_lr_t(++step);
    }
    void adam_range(const Ctx& c, int64_t lo, int64_t hi) {
        if (hi <= lo) return;
        const float lr_t = cur_lr_t;
        launch(c, "adam_kernel", 0, 28.0 * (hi - lo), [&]() {
            return p3d_adam(flat_p + lo, flat_g + lo, flat_m + lo, flat_v + lo, hi - lo, lr_t, c.lr_dev, b1, b2, eps, c.s);
        });
    }
    void run_adam(const Ctx& c) { adam_begin(c); adam_range(c, 0, n_train); }

    // ---- captured train step (opt-in: P3D_GRAPH=1) ---------------------------------------------------
    // One train step is ~1000 dependent launches on three streams.  The launch list is static, so it CAN be captured
    // once into a hipGraph (per dropout rate / pointwise mode / communicator) and replayed; the two per-step scalars
    // (dropout seed, Adam's bias-corrected step size) then live in device memory and are written by a one-thread
    // kernel ahead of each replay.  Measured on MI355X / ROCm 7.2 (profiles/r02_graph_vs_eager.json): a replay costs the
    // host 17.1 ms per step against 9.2 ms of eager enqueueing, and the step takes 20.3 ms instead of 18.2 -- this
    // runtime walks a graph node by node and pays more per kernel node than per eager launch, so replay is slower
    // here.  The capture path is kept (and tested) for runtimes where that changes; eager is the default.
    hipGraph_t step_graph = nullptr;
    hipGraphExec_t step_exec = nullptr;
    float graph_drop = -1.f; bool graph_f16 = false; ncclComm_t graph_comm = nullptr; float graph_b1 = 0, graph_b2 = 0, graph_eps = 0;
    bool graph_disabled = false;
    unsigned long long* d_seed = nullptr; float* d_lr = nullptr;
    void drop_step_graph() {
        if (step_exec) { hipGraphExecDestroy(step_exec); step_exec = nullptr; }
        if (step_graph) { hipGraphDestroy(step_graph); step_graph = nullptr; }
    }
    bool graphs_enabled() {
        return runtime_env().graph && !graph_disabled;
    }
    void capture_step_graph(float drop) {
        drop_step_graph();
        Ctx c; c.training = true; c.drop = drop; c.seed = 0; c.seed_dev = d_seed; c.lr_dev = d_lr; c.update_moving = true; c.s = stream;
        HIPCHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
        try {
            run_forward(c);
            run_loss(c);
            if (!run_backward(c, true, true)) run_adam(c);
        } catch (...) {
            hipGraph_t g = nullptr;
            hipStreamEndCapture(stream, &g);
            if (g) hipGraphDestroy(g);
            hipGetLastError();
            throw;
        }
        HIPCHECK(hipStreamEndCapture(stream, &step_graph));
        HIPCHECK(hipGraphInstantiate(&step_exec, step_graph, nullptr, nullptr, 0));
        graph_drop = drop; graph_f16 = pointwise_f16; graph_comm = comm; graph_b1 = b1; graph_b2 = b2; graph_eps = eps;
    }
    void train_step_device(float drop, uint64_t seed) {
        if (!graphs_enabled()) {
            Ctx c; c.training = true; c.drop = drop; c.seed = seed; c.update_moving = true; c.s = stream;
            zero_early(c);
            run_forward(c); run_loss(c);
            if (!run_backward(c, true, true)) run_adam(c);
            return;
        }
        if (!step_exec || graph_drop != drop || graph_f16 != pointwise_f16 || graph_comm != comm || graph_b1 != b1 || graph_b2 != b2 ||
            graph_eps != eps) {
            try {
                capture_step_graph(drop);
            } catch (const std::exception& e) {
                // a runtime that cannot capture this launch list (e.g. a collective that refuses capture) still trains
                fprintf(stderr, "[p3d] step graph capture failed (%s); using the eager launch list\n", e.what());
                graph_disabled = true;
                drop_step_graph();
                train_step_device(drop, seed);
                return;
            }
        }
        HIPCHECK(p3d_set_step_scalars(d_seed, d_lr, seed, adam_lr_t(++step), stream));
        HIPCHECK(hipGraphLaunch(step_exec, stream));
    }

    void upload(const float* x, const float* y) {
        if (x) HIPCHECK(hipMemcpyAsync(x_in->p, x, (size_t)x_in->rows() * 3 * sizeof(float), hipMemcpyHostToDevice, stream));
        if (y) HIPCHECK(hipMemcpyAsync(d_y, y, (size_t)pred->rows() * sizeof(float), hipMemcpyHostToDevice, stream));
    }
    float read_loss() {
        double l = 0;
        HIPCHECK(hipMemcpyAsync(&l, d_loss, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        return (float)l;
    }
    void download_act(Act* a, float* host) {
        if (a->ld == a->C)
            HIPCHECK(hipMemcpyAsync(host, a->p, (size_t)a->rows() * a->C * sizeof(float), hipMemcpyDeviceToHost, stream));
        else
            HIPCHECK(hipMemcpy2DAsync(host, (size_t)a->C * 4, a->p, (size_t)a->ld * 4, (size_t)a->C * 4, (size_t)a->rows(),
                                      hipMemcpyDeviceToHost, stream));
        HIPCHECK(hipStreamSynchronize(stream));
    }

    ~p3d_handle() {
        drop_step_graph();
        if (comm) ncclCommDestroy(comm);
        if (ev_bucket) hipEventDestroy(ev_bucket);
        if (ev_comm_done) hipEventDestroy(ev_comm_done);
        for (hipEvent_t e : fork_events) hipEventDestroy(e);
        for (hipEvent_t e : wq_events) hipEventDestroy(e);
        if (ev_zeroed) hipEventDestroy(ev_zeroed);
        if (ev_zero_fork) hipEventDestroy(ev_zero_fork);
        if (ev_side_done) hipEventDestroy(ev_side_done);
        if (ev_side_early) hipEventDestroy(ev_side_early);
        if (ev_comm_early) hipEventDestroy(ev_comm_early);
        if (ev_side_bucket) hipEventDestroy(ev_side_bucket);
        if (side_stream) hipStreamDestroy(side_stream);
        for (void* p : allocs) hipFree(p);
        if (comm_stream) hipStreamDestroy(comm_stream);
        if (stream) hipStreamDestroy(stream);
    }
};

// ==================================================================================================
#define API_BEGIN try {
#define API_END                                        \
    }                                                  \
    catch (const std::exception& e) {                  \
        g_err = e.what();                              \
        return -1;                                     \
    }                                                  \
    return 0;

extern "C" {

const char* p3d_last_error(void) { return g_err.c_str(); }

void p3d_default_config(p3d_config* c) {
    memset(c, 0, sizeof(*c));
    c->structure = P3D_STRUCTURE_UNET;
    c->batch = 2; c->frames = 16; c->height = 112; c->width = 112; c->base = 64;
    c->blocks[0] = 3; c->blocks[1] = 8; c->blocks[2] = 36;
    c->device = 0; c->world_size = 1; c->rank = 0;
}

// Two HIP runtimes in one process (e.g. /opt/rocm's, which this library links, next to the copy a PyTorch wheel bundles)
// end in heap corruption at exit.  Callers that bind the library by hand get the diagnosis here instead of there.
static void refuse_two_hip_runtimes() {
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return;
    std::vector<std::string> seen;
    char line[1024];
    while (fgets(line, sizeof(line), f)) {
        const char* p = strstr(line, "libamdhip64.so");
        if (!p) continue;
        const char* path = strchr(line, '/');
        if (!path) continue;
        std::string s(path);
        while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
        if (std::find(seen.begin(), seen.end(), s) == seen.end()) seen.push_back(s);
    }
    fclose(f);
    if (seen.size() > 1)
        throw P3dError("two HIP runtimes are mapped into this process (" + seen[0] + " and " + seen[1] +
                       "): load the one PyTorch bundles before libp3dhip.so (INTEGRATION.md, 'One HIP runtime per process')");
}

int p3d_create(const p3d_config* cfg, p3d_handle** out) {
    p3d_handle* h = nullptr;
    try {
        if (!cfg || !out) throw P3dError("null argument");
        refuse_two_hip_runtimes();
        int ndev = 0;
        HIPCHECK(hipGetDeviceCount(&ndev));
        if (ndev <= 0) throw P3dError("no HIP device: libp3dhip has no CPU fallback");
        if (cfg->device < 0 || cfg->device >= ndev) throw P3dError("bad device ordinal");
        HIPCHECK(hipSetDevice(cfg->device));
        h = new p3d_handle();
        h->cfg = *cfg;
        if (const long mb = bucket_mb_env(); mb >= 1) h->bucket_floats = (int64_t)mb * (1 << 18);
        if (const char* e = p3d_tune_env("P3D_FUSE_MAX_ROWS")) h->fuse_max_rows = atoll(e);      // A/B runs: which bottlenecks are built fusable
        ensure_zero_page();
        {   // the main stream carries the dependent chain of small launches and the comm stream the all-reduces: both above
            // the side stream's filter gradients (measured: no effect on the step time on this ROCm, 17.98 vs 17.95 ms;
            // what does help is keeping the filter gradients' residency low, conv_wgrad2.hip launch_group_t)
            int least = 0, greatest = 0;
            HIPCHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHECK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, greatest));
            HIPCHECK(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, greatest));
            HIPCHECK(hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, least));
        }
        HIPCHECK(hipEventCreateWithFlags(&h->ev_side_done, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&h->ev_side_early, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&h->ev_comm_early, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&h->ev_side_bucket, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&h->ev_bucket, hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&h->ev_comm_done, hipEventDisableTiming));
        if (cfg->structure < P3D_STRUCTURE_UNET || cfg->structure > P3D_STRUCTURE_UNETPP_DS) throw P3dError("unknown structure");
        if (cfg->batch < 1) throw P3dError("batch must be >= 1");
        for (int i = 0; i < 3; ++i)
            if (cfg->blocks[i] < 1) throw P3dError("blocks must be >= 1");
        if (cfg->structure == P3D_STRUCTURE_CONCAT) h->build_concat();
        else if (cfg->structure == P3D_STRUCTURE_GN_P3D) h->build_gn_p3d(16);
        else if (cfg->structure == P3D_STRUCTURE_GN_P3D_CONCAT) h->build_gn_p3d(8);
        else if (cfg->structure == P3D_STRUCTURE_UNETPP_NONSA) h->build_unetpp(false);
        else if (cfg->structure == P3D_STRUCTURE_UNETPP_DS) h->build_unetpp(true);
        else if (cfg->structure == P3D_STRUCTURE_GN_P3D_DECODER) h->build_gn_decoder();
        else h->build_unet();
        h->finalize_build();
        HIPCHECK(hipStreamSynchronize(h->stream));
        *out = h;
        ++g_live_handles;
    } catch (const std::exception& e) {
        g_err = e.what();
        delete h;
        return -1;
    }
    return 0;
}

void p3d_destroy(p3d_handle* h) {
    if (!h) return;
    hipSetDevice(h->cfg.device);
    hipDeviceSynchronize();
    delete h;
    --g_live_handles;
}

int p3d_num_params(p3d_handle* h) { return h ? (int)h->porder.size() : -1; }

int p3d_param_info(p3d_handle* h, int index, const char** name, int* ndim, int64_t shape[5], int* trainable) {
    API_BEGIN
    if (!h || index < 0 || index >= (int)h->porder.size()) throw P3dError("bad parameter index");
    Param* p = h->porder[index];
    if (name) *name = p->name.c_str();
    if (ndim) *ndim = (int)p->shape.size();
    if (shape)
        for (size_t i = 0; i < 5; ++i) shape[i] = i < p->shape.size() ? p->shape[i] : 1;
    if (trainable) *trainable = p->trainable;
    API_END
}

static Param* find_param(p3d_handle* h, const char* name, int64_t count) {
    if (!h || !name) throw P3dError("null argument");
    auto it = h->pindex.find(name);
    if (it == h->pindex.end()) throw P3dError(std::string("no variable named ") + name);
    if (count != it->second->count)
        throw P3dError(std::string("size mismatch for ") + name + ": got " + std::to_string(count) + ", variable has " +
                       std::to_string(it->second->count));
    return it->second;
}

int p3d_set_param(p3d_handle* h, const char* name, const float* host, int64_t count) {
    API_BEGIN
    Param* p = find_param(h, name, count);
    HIPCHECK(hipSetDevice(h->cfg.device));
    // the handle's streams are non-blocking: nothing else orders this copy after a step that is still running
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipStreamSynchronize(h->side_stream));
    HIPCHECK(hipStreamSynchronize(h->comm_stream));
    HIPCHECK(hipMemcpy(p->p, host, (size_t)count * 4, hipMemcpyHostToDevice));
    API_END
}
int p3d_get_param(p3d_handle* h, const char* name, float* host, int64_t count) {
    API_BEGIN
    Param* p = find_param(h, name, count);
    HIPCHECK(hipSetDevice(h->cfg.device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipMemcpy(host, p->p, (size_t)count * 4, hipMemcpyDeviceToHost));
    API_END
}
int p3d_get_grad(p3d_handle* h, const char* name, float* host, int64_t count) {
    API_BEGIN
    Param* p = find_param(h, name, count);
    if (!p->trainable) throw P3dError(std::string(name) + " is not trainable");
    HIPCHECK(hipSetDevice(h->cfg.device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipMemcpy(host, p->g, (size_t)count * 4, hipMemcpyDeviceToHost));
    API_END
}

int p3d_init_params(p3d_handle* h, uint64_t seed) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    uint64_t idx = 0;
    for (Param* p : h->porder) {
        ++idx;
        if (p->init == INIT_XAVIER) {
            // tf.contrib.layers.xavier_initializer: U(-L, L), L = sqrt(6 / (fan_in + fan_out)); rank-1 [C]: fans = C
            double fi, fo;
            if (p->shape.size() == 1) fi = fo = (double)p->shape[0];
            else {
                double rf = 1;
                for (size_t i = 0; i + 2 < p->shape.size(); ++i) rf *= (double)p->shape[i];
                fi = rf * p->shape[p->shape.size() - 2];
                fo = rf * p->shape[p->shape.size() - 1];
            }
            const float L = (float)std::sqrt(6.0 / (fi + fo));
            HIPCHECK(p3d_fill_uniform(p->p, p->count, -L, L, seed * 0x9E3779B97F4A7C15ull + idx, h->stream));
        } else if (p->init == INIT_VS) {
            // tf.contrib.layers.variance_scaling_initializer() defaults (utils/network.py:212-213,264): factor 2.0,
            // mode FAN_IN, uniform False -> truncated normal with stddev sqrt(1.3 * 2 / fan_in) (the 1.3 is TF's
            // correction for the variance the truncation at two standard deviations removes)
            double rf = 1;
            for (size_t i = 0; i + 2 < p->shape.size(); ++i) rf *= (double)p->shape[i];
            const double fan_in = rf * p->shape[p->shape.size() - 2];
            HIPCHECK(p3d_fill_trunc_normal(p->p, p->count, (float)std::sqrt(1.3 * 2.0 / fan_in), seed * 0x9E3779B97F4A7C15ull + idx, h->stream));
        } else {
            const float v = p->init == INIT_ONES ? 1.f : 0.f;
            HIPCHECK(p3d_fill_uniform(p->p, p->count, v, v, 0, h->stream));
        }
    }
    HIPCHECK(hipMemsetAsync(h->flat_m, 0, (size_t)h->n_train * 4, h->stream));
    HIPCHECK(hipMemsetAsync(h->flat_v, 0, (size_t)h->n_train * 4, h->stream));
    h->step = 0;
    HIPCHECK(hipStreamSynchronize(h->stream));
    API_END
}

int p3d_upload_inputs(p3d_handle* h, const float* x, const float* y) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    h->upload(x, y);
    HIPCHECK(hipStreamSynchronize(h->stream));
    API_END
}

int p3d_forward_device(p3d_handle* h, int training, float dropout_rate, uint64_t seed) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    Ctx c; c.training = training != 0; c.drop = dropout_rate; c.seed = seed; c.update_moving = false; c.s = h->stream;
    h->run_forward(c);
    API_END
}

int p3d_forward(p3d_handle* h, const float* x, int training, float dropout_rate, uint64_t seed, float* pred) {
    API_BEGIN
    if (!h || !x || !pred) throw P3dError("null argument");
    HIPCHECK(hipSetDevice(h->cfg.device));
    h->upload(x, nullptr);
    Ctx c; c.training = training != 0; c.drop = dropout_rate; c.seed = seed; c.update_moving = false; c.s = h->stream;
    h->run_forward(c);
    h->download_act(h->pred, pred);
    API_END
}

int p3d_set_pointwise_fp16(p3d_handle* h, int enable) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    h->pointwise_f16 = enable != 0;
    API_END
}

int64_t p3d_debug_dirty_counters(void) { return (int64_t)p3d_scratch_dirty_counters(); }

int p3d_debug_force_plan(int igemm_tile, int igemm_splits, int wgrad_tm, int wgrad_tn) {
    p3d_igemm2_override(igemm_tile, igemm_splits);
    p3d_wgrad2_force_tile(wgrad_tm, wgrad_tn);
    return 0;
}

int p3d_set_attention_mode(p3d_handle* h, int mode) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    if (mode < 0 || mode > 2) throw P3dError("attention mode is 0 (per site), 1 (stored scores) or 2 (flash)");
    HIPCHECK(hipSetDevice(h->cfg.device));
    if (mode == 1) for (auto& f : h->attn_gemm_alloc) f();      // sites that were built flash-only get their score buffers now
    h->attn_mode = mode;
    h->drop_step_graph();
    API_END
}

int p3d_set_bn_fusion(p3d_handle* h, int enable) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    h->fuse_bn = enable != 0;
    h->fuse_bn_bwd = enable >= 2;
    h->drop_step_graph();
    API_END
}

int p3d_predict_windows(p3d_handle* h, const float* x, float* pred) {
    API_BEGIN
    if (!h || !x || !pred) throw P3dError("null argument");
    HIPCHECK(hipSetDevice(h->cfg.device));
    h->upload(x, nullptr);
    Ctx c; c.training = false; c.drop = 0.f; c.update_moving = false; c.per_sample = true; c.s = h->stream;
    h->run_forward(c);
    h->download_act(h->pred, pred);
    API_END
}

int p3d_backward(p3d_handle* h, const float* x, const float* y, float dropout_rate, uint64_t seed, float* loss, float* pred) {
    API_BEGIN
    if (!h || !x || !y) throw P3dError("null argument");
    HIPCHECK(hipSetDevice(h->cfg.device));
    h->upload(x, y);
    Ctx c; c.training = true; c.drop = dropout_rate; c.seed = seed; c.update_moving = false; c.s = h->stream;
    h->run_forward(c);
    h->run_loss(c);
    h->run_backward(c, false);
    const float l = h->read_loss();
    if (loss) *loss = l;
    if (pred) h->download_act(h->pred, pred);
    API_END
}

int p3d_train_step_device(p3d_handle* h, float dropout_rate, uint64_t seed) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    h->train_step_device(dropout_rate, seed);
    API_END
}

int p3d_train_step(p3d_handle* h, const float* x, const float* y, float dropout_rate, uint64_t seed, float* loss) {
    API_BEGIN
    if (!h || !x || !y) throw P3dError("null argument");
    HIPCHECK(hipSetDevice(h->cfg.device));
    h->upload(x, y);
    h->train_step_device(dropout_rate, seed);
    const float l = h->read_loss();
    if (loss) *loss = l;
    API_END
}

int p3d_last_loss(p3d_handle* h, float* loss) {
    API_BEGIN
    if (!h || !loss) throw P3dError("null argument");
    HIPCHECK(hipSetDevice(h->cfg.device));
    *loss = h->read_loss();
    API_END
}

int p3d_synchronize(p3d_handle* h) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipStreamSynchronize(h->side_stream));
    HIPCHECK(hipStreamSynchronize(h->comm_stream));
    API_END
}

int p3d_set_adam(p3d_handle* h, float lr, float beta1, float beta2, float eps) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    h->lr = lr; h->b1 = beta1; h->b2 = beta2; h->eps = eps;
    API_END
}

int p3d_activation_info(p3d_handle* h, const char* name, int64_t shape[5]) {
    API_BEGIN
    if (!h || !name) throw P3dError("null argument");
    auto it = h->named.find(name);
    if (it == h->named.end()) throw P3dError(std::string("no activation named ") + name);
    Act* a = it->second;
    shape[0] = a->N; shape[1] = a->D; shape[2] = a->H; shape[3] = a->W; shape[4] = a->C;
    API_END
}

int p3d_get_activation(p3d_handle* h, const char* name, float* host, int64_t count) {
    API_BEGIN
    if (!h || !name || !host) throw P3dError("null argument");
    auto it = h->named.find(name);
    if (it == h->named.end()) throw P3dError(std::string("no activation named ") + name);
    Act* a = it->second;
    if (count != a->rows() * a->C) throw P3dError("activation size mismatch");
    HIPCHECK(hipSetDevice(h->cfg.device));
    if (a->materialize && h->last_forward_fused) a->materialize(h->stream);
    h->download_act(a, host);
    API_END
}

int p3d_block_info(p3d_handle* h, int block_id, int64_t in_shape[5], int64_t out_shape[5]) {
    API_BEGIN
    if (!h) throw P3dError("null handle");
    auto it = h->blocks.find(block_id);
    if (it == h->blocks.end()) throw P3dError("no bottleneck with id " + std::to_string(block_id));
    const Act* a = it->second.in; const Act* b = it->second.out;
    if (in_shape) { in_shape[0] = a->N; in_shape[1] = a->D; in_shape[2] = a->H; in_shape[3] = a->W; in_shape[4] = a->C; }
    if (out_shape) { out_shape[0] = b->N; out_shape[1] = b->D; out_shape[2] = b->H; out_shape[3] = b->W; out_shape[4] = b->C; }
    API_END
}

int p3d_block_forward(p3d_handle* h, int block_id, const float* in, int64_t in_count, float* out, int64_t out_count) {
    API_BEGIN
    if (!h || !in || !out) throw P3dError("null argument");
    auto it = h->blocks.find(block_id);
    if (it == h->blocks.end()) throw P3dError("no bottleneck with id " + std::to_string(block_id));
    Act* a = it->second.in; Act* b = it->second.out;
    if (in_count != a->rows() * a->C || out_count != b->rows() * b->C) throw P3dError("block tensor size mismatch");
    HIPCHECK(hipSetDevice(h->cfg.device));
    HIPCHECK(hipMemcpy2DAsync(a->p, (size_t)a->ld * 4, in, (size_t)a->C * 4, (size_t)a->C * 4, (size_t)a->rows(), hipMemcpyHostToDevice, h->stream));
    Ctx c; c.training = true; c.s = h->stream; c.fuse = h->fuse_bn;
    h->last_forward_fused = c.fuse;
    h->sib_pending.clear();
    if (h->stats_count) HIPCHECK(hipMemsetAsync(h->stats_arena, 0, (size_t)h->stats_count * sizeof(double), c.s));
    for (size_t i = it->second.op0; i < it->second.op1; ++i) h->ops[i].fwd(c);      // (no zero arena: ops zero what they slice)
    h->download_act(b, out);
    API_END
}

int p3d_block_backward(p3d_handle* h, int block_id, const float* in, int64_t in_count, const float* dout, int64_t out_count, float* din) {
    API_BEGIN
    if (!h || !in || !dout || !din) throw P3dError("null argument");
    auto it = h->blocks.find(block_id);
    if (it == h->blocks.end()) throw P3dError("no bottleneck with id " + std::to_string(block_id));
    Act* a = it->second.in; Act* b = it->second.out;
    if (in_count != a->rows() * a->C || out_count != b->rows() * b->C) throw P3dError("block tensor size mismatch");
    if (!a->g || !b->g) throw P3dError("this bottleneck's input or output carries no gradient");
    HIPCHECK(hipSetDevice(h->cfg.device));
    HIPCHECK(hipMemcpy2DAsync(a->p, (size_t)a->ld * 4, in, (size_t)a->C * 4, (size_t)a->C * 4, (size_t)a->rows(), hipMemcpyHostToDevice, h->stream));
    Ctx c; c.training = true; c.s = h->stream; c.fuse = false;
    h->sib_pending.clear();
    h->last_forward_fused = false;
    if (h->stats_count) HIPCHECK(hipMemsetAsync(h->stats_arena, 0, (size_t)h->stats_count * sizeof(double), c.s));
    for (size_t i = it->second.op0; i < it->second.op1; ++i) h->ops[i].fwd(c);
    // the backward walk of run_backward over this block's ops only, on one stream; the block's output gradient is given
    h->zeroed_early = false;
    h->zero_backward_arenas(c.s, true);
    c.z0 = h->zb; c.z1 = h->zb + h->zb_bytes;
    HIPCHECK(hipMemcpy2DAsync(b->g, (size_t)b->ld * 4, dout, (size_t)b->C * 4, (size_t)b->C * 4, (size_t)b->rows(), hipMemcpyHostToDevice, c.s));
    h->wq.clear(); h->wq_flushes = 0; h->parked_flops = 0;
    for (size_t i = it->second.op1; i-- > it->second.op0;) {
        c.bwd_op = h->ops[i].name.c_str();
        h->ops[i].bwd(c);
    }
    h->flush_wgrads(c);
    HIPCHECK(hipStreamSynchronize(c.s));
    HIPCHECK(hipMemcpy2D(din, (size_t)a->C * 4, a->g, (size_t)a->ld * 4, (size_t)a->C * 4, (size_t)a->rows(), hipMemcpyDeviceToHost));
    API_END
}

int p3d_profile_step(p3d_handle* h, float dropout_rate, uint64_t seed, p3d_op_time* out, int cap) {
    if (!h) { g_err = "null handle"; return -1; }
    try {
        HIPCHECK(hipSetDevice(h->cfg.device));
        Prof prof;
        Ctx c; c.training = true; c.drop = dropout_rate; c.seed = seed; c.update_moving = true; c.s = h->stream; c.prof = &prof;
        prof.phase = 0; h->run_forward(c);
        prof.cur_op = "loss"; h->run_loss(c);
        // no collective here: bench.py profiles on rank 0 only, after the timed region -- an all-reduce that the
        // other ranks do not enter would never return
        prof.phase = 1; h->run_backward(c, false);
        prof.phase = 2; prof.cur_op = "adam"; h->run_adam(c);
        HIPCHECK(hipStreamSynchronize(c.s));
        int w = 0;
        for (auto& r : prof.recs) {
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
            if (w < cap && out) {
                p3d_op_time& o = out[w];
                memset(&o, 0, sizeof(o));
                snprintf(o.name, sizeof(o.name), "%s", r.op.c_str());
                snprintf(o.kernel, sizeof(o.kernel), "%s", r.kernel.c_str());
                o.ms = ms; o.flops = r.flops; o.bytes = r.bytes; o.phase = r.phase;
            }
            ++w;
            hipEventDestroy(r.e0);
            hipEventDestroy(r.e1);
        }
        return w;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

int p3d_debug_bucket_audit(p3d_handle* h, float dropout_rate, uint64_t seed, int64_t bucket_floats, int64_t* lo, int64_t* hi,
                           int32_t* after_op, int cap, int64_t* n_train, int64_t* stale) {
    if (!h) { g_err = "null handle"; return -1; }
    int count = 0;
    try {
        HIPCHECK(hipSetDevice(h->cfg.device));
        if (bucket_floats < 1) throw P3dError("bucket size must be positive");
        struct Snap { int64_t lo, hi; int op; std::vector<float> g; };
        std::vector<Snap> snaps;
        const int64_t saved = h->bucket_floats;
        h->bucket_floats = bucket_floats;
        h->bucket_hook = [&](int64_t l, int64_t u, int op) {
            // everything the bucket's gradients depend on must already be QUEUED: wait for it, then look
            HIPCHECK(hipStreamSynchronize(h->stream));
            HIPCHECK(hipStreamSynchronize(h->side_stream));
            Snap s; s.lo = l; s.hi = u; s.op = op; s.g.resize((size_t)(u - l));
            HIPCHECK(hipMemcpy(s.g.data(), h->flat_g + l, (size_t)(u - l) * 4, hipMemcpyDeviceToHost));
            snaps.push_back(std::move(s));
        };
        try {
            Ctx c; c.training = true; c.drop = dropout_rate; c.seed = seed; c.update_moving = false; c.s = h->stream;
            h->run_forward(c);
            h->run_loss(c);
            h->run_backward(c, true);
            HIPCHECK(hipStreamSynchronize(h->stream));
            HIPCHECK(hipStreamSynchronize(h->side_stream));
        } catch (...) {
            h->bucket_hook = nullptr; h->bucket_floats = saved;
            throw;
        }
        h->bucket_hook = nullptr; h->bucket_floats = saved;
        std::vector<float> fin((size_t)h->n_train);
        HIPCHECK(hipMemcpy(fin.data(), h->flat_g, (size_t)h->n_train * 4, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (auto& s : snaps) {
            if (memcmp(s.g.data(), fin.data() + s.lo, s.g.size() * 4) != 0)
                for (size_t i = 0; i < s.g.size(); ++i)
                    if (memcmp(&s.g[i], &fin[(size_t)s.lo + i], 4) != 0) ++bad;
            if (count < cap) {
                if (lo) lo[count] = s.lo;
                if (hi) hi[count] = s.hi;
                if (after_op) after_op[count] = s.op;
            }
            ++count;
        }
        if (n_train) *n_train = h->n_train;
        if (stale) *stale = bad;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
    return count;
}

int p3d_comm_unique_id(void* id_out) {
    API_BEGIN
    static_assert(sizeof(ncclUniqueId) <= P3D_COMM_ID_BYTES, "id size");
    ncclUniqueId id;
    NCCLCHECK(ncclGetUniqueId(&id));
    memset(id_out, 0, P3D_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    API_END
}

int p3d_comm_init(p3d_handle* h, const void* idbytes) {
    API_BEGIN
    if (!h || !idbytes) throw P3dError("null argument");
    if (h->cfg.world_size < 1) throw P3dError("bad world_size");      // world_size 1 is allowed (single-rank communicator, for tests)
    HIPCHECK(hipSetDevice(h->cfg.device));
    ncclUniqueId id;
    memcpy(&id, idbytes, sizeof(id));
    NCCLCHECK(ncclCommInitRank(&h->comm, h->cfg.world_size, id, h->cfg.rank));
    if (const long mb = bucket_mb_env(); mb >= 1) h->bucket_floats = (int64_t)mb * (1 << 18);
    API_END
}

// ---- single-operator entry points ------------------------------------------------------------------
namespace {
struct DevBuf {
    float* p = nullptr;
    explicit DevBuf(int64_t n, const float* host = nullptr) {
        HIPCHECK(hipMalloc((void**)&p, (size_t)(n > 0 ? n : 1) * 4));
        if (host) HIPCHECK(hipMemcpy(p, host, (size_t)n * 4, hipMemcpyHostToDevice));
        else HIPCHECK(hipMemset(p, 0, (size_t)(n > 0 ? n : 1) * 4));
    }
    ~DevBuf() { hipFree(p); }
    void get(float* host, int64_t n) { HIPCHECK(hipDeviceSynchronize()); HIPCHECK(hipMemcpy(host, p, (size_t)n * 4, hipMemcpyDeviceToHost)); }
};
int64_t prod5(const int64_t s[5]) { return s[0] * s[1] * s[2] * s[3] * s[4]; }
bool is_stem_shape(const int64_t xs[5], const int64_t ws[5]) { return xs[4] % 4 != 0 && ws[0] == 1; }
}  // namespace

// ---- decisions of the last forward (test hook, include/p3d_hip.h) ---------------------------------------------------------
static const Op* decision_op(p3d_handle* h, int index) {
    if (!h) throw P3dError("null handle");
    int k = 0;
    for (const Op& op : h->ops)
        if (!op.dec_kind.empty() && k++ == index) return &op;
    throw P3dError("no decision site " + std::to_string(index));
}
int p3d_debug_decision_count(p3d_handle* h) {
    if (!h) return -1;
    int k = 0;
    for (const Op& op : h->ops) k += !op.dec_kind.empty();
    return k;
}
int p3d_debug_decision_info(p3d_handle* h, int index, const char** kind, const char** name1, const char** name2, int64_t shape[5]) {
    API_BEGIN
    const Op* op = decision_op(h, index);
    if (kind) *kind = op->dec_kind.c_str();
    if (name1) *name1 = op->dec_name1.c_str();
    if (name2) *name2 = op->dec_name2.c_str();
    const Act* a = op->dec_act;
    if (shape) { shape[0] = a->N; shape[1] = a->D; shape[2] = a->H; shape[3] = a->W; shape[4] = a->C; }
    API_END
}
int p3d_debug_decision_get(p3d_handle* h, int index, float* out1, float* out2, int64_t count) {
    API_BEGIN
    const Op* op = decision_op(h, index);
    const Act* a = op->dec_act;
    if (!out1 || count != a->rows() * a->C) throw P3dError("decision buffer size mismatch");
    HIPCHECK(hipSetDevice(h->cfg.device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (op->dec_kind == "pool") {
        h->download_act(const_cast<Act*>(a), out1);
    } else {
        if (h->last_forward_fused) throw P3dError("decisions are read from the stored tensors: run the forward with BatchNorm fusion off");
        std::vector<float> ones((size_t)count, 1.0f);
        DevBuf d1(count, ones.data()), o1(count), o2(count), scratch(4 * (int64_t)a->C + 64);
        op->gates(h->stream, d1.p, o1.p, o2.p, scratch.p);
        o1.get(out1, count);
        if (out2) o2.get(out2, count);
    }
    API_END
}

int p3d_op_conv3d(int device, const float* x, const int64_t xs[5], const float* w, const int64_t ws[5], const int s[3],
                  const float* bias, float* y) {
    API_BEGIN
    HIPCHECK(hipSetDevice(device));
    const int k[3] = {(int)ws[0], (int)ws[1], (int)ws[2]};
    const ConvGeo g = make_geo((int)xs[1], (int)xs[2], (int)xs[3], k, s);
    const int Cin = (int)xs[4], Cout = (int)ws[4];
    const int64_t ny = xs[0] * g.O[0] * g.O[1] * g.O[2] * Cout;
    DevBuf dx(prod5(xs), x), dw(prod5(ws), w), dy(ny), db(Cout, bias);
    ensure_zero_page();
    Ctx c;
    if (is_stem_shape(xs, ws)) {          // [1,kh,kw,3,Cout] on its packed form, like the network's stem
        if (Cin != 3) throw P3dError("conv3d: channel counts that are not multiples of 4 are supported for the 3-channel stem only");
        const StemGeo sg = stem_geo(g, (int)xs[0]);
        DevBuf x4(sg.xrows * sg.Wp * 4), w4((int64_t)sg.KH * sg.K4 * Cout);
        HIPCHECK(p3d_stem_pad(dx.p, x4.p, sg.xrows, g.I[2], sg.Wp, g.pad[2], c.s));
        HIPCHECK(p3d_stem_pack_w(dw.p, w4.p, sg.KH * g.k[2], Cout, c.s));
        std::vector<IgemmArgs> v{stem_forward_args(g, (int)xs[0], sg, x4.p, w4.p, dy.p, Cout, Cout, bias ? db.p : nullptr)};
        run_igemm_group(c, v, dy.p, Cout, ny / Cout, Cout, false, nullptr);
        HIPCHECK(hipDeviceSynchronize());
    } else {
        std::vector<IgemmArgs> v{igemm_conv_forward(g, (int)xs[0], dx.p, Cin, Cin, dy.p, Cout, Cout, dw.p, bias ? db.p : nullptr, 0, false)};
        run_igemm_group(c, v, dy.p, Cout, ny / Cout, Cout, false, nullptr);
    }
    dy.get(y, ny);
    API_END
}

int p3d_op_conv3d_backprop_input(int device, const float* dyh, const float* w, const int64_t ws[5], const int s[3],
                                 const int64_t xs[5], float* dxh) {
    API_BEGIN
    HIPCHECK(hipSetDevice(device));
    const int k[3] = {(int)ws[0], (int)ws[1], (int)ws[2]};
    const ConvGeo g = make_geo((int)xs[1], (int)xs[2], (int)xs[3], k, s);
    const int Cin = (int)xs[4], Cout = (int)ws[4];
    const int64_t ny = xs[0] * g.O[0] * g.O[1] * g.O[2] * Cout;
    DevBuf dy(ny, dyh), dw(prod5(ws), w), dx(prod5(xs));
    auto v = igemm_conv_input_side(g, (int)xs[0], dy.p, Cout, Cout, dx.p, Cin, Cin, dw.p, nullptr, 0, true);
    ensure_zero_page();
    { Ctx c; run_igemm_group(c, v, dx.p, Cin, prod5(xs) / Cin, Cin, false, nullptr); }
    dx.get(dxh, prod5(xs));
    API_END
}

int p3d_op_conv3d_backprop_filter(int device, const float* x, const int64_t xs[5], const float* dyh, const int64_t ws[5],
                                  const int s[3], float* dwh, float* dbh) {
    API_BEGIN
    HIPCHECK(hipSetDevice(device));
    const int k[3] = {(int)ws[0], (int)ws[1], (int)ws[2]};
    const ConvGeo g = make_geo((int)xs[1], (int)xs[2], (int)xs[3], k, s);
    const int Cin = (int)xs[4], Cout = (int)ws[4];
    const int64_t ny = xs[0] * g.O[0] * g.O[1] * g.O[2] * Cout;
    DevBuf dx(prod5(xs), x), dy(ny, dyh), dw(prod5(ws)), db(Cout);
    ensure_zero_page();
    if (is_stem_shape(xs, ws)) {          // [1,kh,kw,3,Cout] on its packed form, like the network's stem (no atomics anywhere)
        if (Cin != 3) throw P3dError("conv3d_backprop_filter: channel counts that are not multiples of 4 are supported for the 3-channel stem only");
        const StemGeo sg = stem_geo(g, (int)xs[0]);
        const int Wp = sg.Wp;
        const int64_t xrows = sg.xrows;
        DevBuf x4(xrows * Wp * 4), dw4((int64_t)sg.KH * sg.K4 * Cout);
        Ctx c;
        HIPCHECK(p3d_stem_pad(dx.p, x4.p, xrows, g.I[2], Wp, g.pad[2], c.s));
        stem_filter_gradient(c, g, (int)xs[0], Wp, x4.p, dy.p, Cout, Cout, dw4.p, dw.p, dbh ? db.p : nullptr, false);
        HIPCHECK(hipDeviceSynchronize());
    } else {
        WgradArgs a = wgrad_conv(g, (int)xs[0], dx.p, Cin, Cin, dy.p, Cout, Cout, dw.p, dbh ? db.p : nullptr, false);
        Ctx c; launch_wgrad(c, a);
    }
    dw.get(dwh, prod5(ws));
    if (dbh) db.get(dbh, Cout);
    API_END
}

int p3d_op_conv3d_transpose(int device, const float* x, const int64_t xs[5], const float* kh, const int64_t ks[5],
                            const int s[3], const float* bias, float* y) {
    API_BEGIN
    HIPCHECK(hipSetDevice(device));
    const int k[3] = {(int)ks[0], (int)ks[1], (int)ks[2]};
    const ConvGeo g = make_geo((int)xs[1] * s[0], (int)xs[2] * s[1], (int)xs[3] * s[2], k, s);
    const int Cin = (int)xs[4], Cout = (int)ks[3];
    if (ks[4] != Cin) throw P3dError("kernel Cin mismatch");
    const int64_t ny = xs[0] * g.I[0] * g.I[1] * g.I[2] * Cout;
    DevBuf dx(prod5(xs), x), dk(prod5(ks), kh), dy(ny), db(Cout, bias);
    auto v = igemm_conv_input_side(g, (int)xs[0], dx.p, Cin, Cin, dy.p, Cout, Cout, dk.p, bias ? db.p : nullptr, 0, true);
    ensure_zero_page();
    { Ctx c; run_igemm_group(c, v, dy.p, Cout, ny / Cout, Cout, false, nullptr); }
    dy.get(y, ny);
    API_END
}

static PoolArgs pool_args(const int64_t xs[5], const int k[3], const int s[3], const ConvGeo& g) {
    PoolArgs a;
    memset(&a, 0, sizeof(a));
    a.N = (int)xs[0]; a.Di = (int)xs[1]; a.Hi = (int)xs[2]; a.Wi = (int)xs[3]; a.C = (int)xs[4]; a.ldx = a.C;
    a.Do = g.O[0]; a.Ho = g.O[1]; a.Wo = g.O[2]; a.ldy = a.C;
    a.kd = k[0]; a.kh = k[1]; a.kw = k[2]; a.sd = s[0]; a.sh = s[1]; a.sw = s[2];
    a.pd = g.pad[0]; a.ph = g.pad[1]; a.pw = g.pad[2];
    a.lddy = a.C; a.lddx = a.C;
    return a;
}

int p3d_op_max_pool3d(int device, const float* x, const int64_t xs[5], const int k[3], const int s[3], float* y) {
    API_BEGIN
    HIPCHECK(hipSetDevice(device));
    const ConvGeo g = make_geo((int)xs[1], (int)xs[2], (int)xs[3], k, s);
    const int64_t ny = xs[0] * g.O[0] * g.O[1] * g.O[2] * xs[4];
    DevBuf dx(prod5(xs), x), dy(ny);
    PoolArgs a = pool_args(xs, k, s, g);
    a.x = dx.p; a.y = dy.p;
    HIPCHECK(p3d_maxpool_fwd(a, nullptr));
    dy.get(y, ny);
    API_END
}

int p3d_op_max_pool3d_grad(int device, const float* x, const int64_t xs[5], const int k[3], const int s[3], const float* dyh,
                           float* dxh) {
    API_BEGIN
    HIPCHECK(hipSetDevice(device));
    const ConvGeo g = make_geo((int)xs[1], (int)xs[2], (int)xs[3], k, s);
    const int64_t ny = xs[0] * g.O[0] * g.O[1] * g.O[2] * xs[4];
    if (xs[4] % 4) throw P3dError("max_pool3d_grad needs a channel count that is a multiple of 4");
    DevBuf dx(prod5(xs), x), dy(ny, dyh), dg(prod5(xs)), yy(ny), tab(ny / 4);
    PoolArgs a = pool_args(xs, k, s, g);
    a.x = dx.p; a.dy = dy.p; a.dx = dg.p; a.y = yy.p;
    const bool disjoint = p3d_maxpool_disjoint(a);
    if (!disjoint) a.idx = reinterpret_cast<unsigned*>(tab.p);
    HIPCHECK(p3d_maxpool_fwd(a, nullptr));       // the backward kernels read the forward's output (disjoint windows: the first
                                                 // cell equal to the maximum) or its arg-max table (overlapping windows: a gather)
    if (disjoint) HIPCHECK(p3d_maxpool_bwd_disjoint(a, 0, nullptr));
    else HIPCHECK(p3d_maxpool_bwd_gather(a, 0, nullptr));
    dg.get(dxh, prod5(xs));
    API_END
}

int p3d_op_bias_add_grad(int device, const float* dyh, int64_t rows, int channels, float* dbias) {
    API_BEGIN
    if (!dyh || !dbias) throw P3dError("null argument");
    if (rows < 0 || channels < 1) throw P3dError("bias_add_grad needs rows >= 0 and channels >= 1");
    HIPCHECK(hipSetDevice(device));
    DevBuf dy(rows * channels, dyh), db(channels);
    if (rows > 0) HIPCHECK(p3d_colsum(dy.p, channels, (long)rows, channels, db.p, nullptr));
    db.get(dbias, channels);
    API_END
}

int p3d_op_attention_core(int device, int batch, int n_g, int n_f, int ch, const float* gh, const float* fh, const float* hh,
                          float* o, const float* d_o, float* dg, float* df, float* dh) {
    API_BEGIN
    if (!gh || !fh || !hh || !o) throw P3dError("null argument");
    if (!p3d_flash_attn_ok(ch)) throw P3dError("attention_core: ch must be 32, 64, 128 or 256");
    if (batch < 1 || n_g < 1 || n_f < 1) throw P3dError("attention_core needs at least one clip, query and key");
    if (d_o && (!dg || !df || !dh)) throw P3dError("attention_core: the backward pass writes dg, df and dh");
    HIPCHECK(hipSetDevice(device));
    const int ci = ch / 8;
    const int64_t ng = (int64_t)batch * n_g, nf = (int64_t)batch * n_f;
    DevBuf g(ng * ci, gh), f(nf * ci, fh), h(nf * ch, hh), out(ng * ch), lse(ng), dsum(ng);
    FlashAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.B = batch; a.Ng = n_g; a.Nf = n_f; a.ch = ch;
    a.g = g.p; a.ldg = ci; a.f = f.p; a.ldf = ci; a.h = h.p; a.ldh = ch; a.o = out.p; a.ldo = ch; a.lse = lse.p;
    HIPCHECK(p3d_flash_attn_fwd(a, nullptr));
    out.get(o, ng * ch);
    if (d_o) {
        DevBuf dout(ng * ch, d_o), gg(ng * ci), gf(nf * ci), gv(nf * ch);
        a.d_o = dout.p; a.lddo = ch; a.dsum = dsum.p;
        a.dg = gg.p; a.lddg = ci; a.df = gf.p; a.lddf = ci; a.dh = gv.p; a.lddh = ch;
        HIPCHECK(p3d_flash_attn_bwd(a, nullptr));
        gg.get(dg, ng * ci); gf.get(df, nf * ci); gv.get(dh, nf * ch);
    }
    API_END
}

}  // extern "C"

// ---- metrics / pre-processing entry points (metrics.hip) -------------------------------------------------------
namespace {
template <typename T>
struct DevArr {
    T* p = nullptr;
    explicit DevArr(size_t n, const T* host = nullptr) {
        HIPCHECK(hipMalloc((void**)&p, (n > 0 ? n : 1) * sizeof(T)));
        if (host) HIPCHECK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    }
    ~DevArr() { hipFree(p); }
    void get(T* host, size_t n) { HIPCHECK(hipDeviceSynchronize()); HIPCHECK(hipMemcpy(host, p, n * sizeof(T), hipMemcpyDeviceToHost)); }
};
void metric_args(int device, const void* a, const void* b, int n_maps, int n_pix, const void* out) {
    if (!a || !b || !out) throw P3dError("null argument");
    if (n_maps < 1 || n_pix < 1) throw P3dError("metrics need at least one map and one pixel");
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw P3dError("no HIP device: libp3dhip has no CPU fallback");
    if (device < 0 || device >= ndev) throw P3dError("bad device ordinal");
    HIPCHECK(hipSetDevice(device));
}
}  // namespace

extern "C" {

int p3d_metric_cc(int device, const float* a, const float* b, int n_maps, int n_pix, double* out) {
    API_BEGIN
    metric_args(device, a, b, n_maps, n_pix, out);
    const size_t n = (size_t)n_maps * n_pix;
    DevArr<float> da(n, a), db(n, b); DevArr<double> dout(n_maps);
    HIPCHECK(p3d_metric_cc(da.p, db.p, n_maps, n_pix, dout.p, nullptr));
    dout.get(out, n_maps);
    API_END
}
int p3d_metric_sim(int device, const float* a, const float* b, int n_maps, int n_pix, double* out) {
    API_BEGIN
    metric_args(device, a, b, n_maps, n_pix, out);
    const size_t n = (size_t)n_maps * n_pix;
    DevArr<float> da(n, a), db(n, b); DevArr<double> dout(n_maps);
    HIPCHECK(p3d_metric_sim(da.p, db.p, n_maps, n_pix, dout.p, nullptr));
    dout.get(out, n_maps);
    API_END
}
int p3d_metric_nss(int device, const float* sal, const float* fix, int n_maps, int n_pix, double* out) {
    API_BEGIN
    metric_args(device, sal, fix, n_maps, n_pix, out);
    const size_t n = (size_t)n_maps * n_pix;
    DevArr<float> da(n, sal), db(n, fix); DevArr<double> dout(n_maps);
    HIPCHECK(p3d_metric_nss(da.p, db.p, n_maps, n_pix, dout.p, nullptr));
    dout.get(out, n_maps);
    API_END
}
int p3d_metric_auc_judd(int device, const float* sal, const float* fix, const float* jitter, int n_maps, int n_pix, double* out) {
    API_BEGIN
    metric_args(device, sal, fix, n_maps, n_pix, out);
    const size_t n = (size_t)n_maps * n_pix;
    const size_t pad = (size_t)p3d_metric_auc_pad(n_pix);
    DevArr<float> da(n, sal), db(n, fix), dj(jitter ? n : 1, jitter), thr(pad * n_maps);
    DevArr<int> cnt((pad + 1) * n_maps);
    DevArr<double> dout(n_maps);
    HIPCHECK(p3d_metric_auc_judd(da.p, db.p, jitter ? dj.p : nullptr, n_maps, n_pix, thr.p, cnt.p, dout.p, nullptr));
    dout.get(out, n_maps);
    API_END
}
int p3d_metric_auc_borji(int device, const float* sal, const float* fix, const int* rand_idx, int n_pix, int n_fix, int n_rep,
                         double step_size, double* out) {
    API_BEGIN
    metric_args(device, sal, fix, 1, n_pix, out);
    if (!rand_idx || n_rep < 1 || !(step_size > 0.0)) throw P3dError("AUC_Borji needs random indices, n_rep >= 1 and a positive step");
    DevArr<float> da(n_pix, sal), db(n_pix, fix);
    DevArr<int> fidx(n_pix), fcount(1);
    HIPCHECK(p3d_metric_fix_index(db.p, n_pix, fidx.p, fcount.p, nullptr));
    int have = 0;
    fcount.get(&have, 1);
    if (have == 0) { for (int i = 0; i < n_rep; ++i) out[i] = NAN; return 0; }      // "no fixation to predict"
    if (have != n_fix) throw P3dError("AUC_Borji: n_fix = " + std::to_string(n_fix) + " but the fixation map has " + std::to_string(have) + " fixated pixels");
    for (size_t i = 0; i < (size_t)n_fix * n_rep; ++i)
        if (rand_idx[i] < 0 || rand_idx[i] >= n_pix) throw P3dError("AUC_Borji: random index out of range");
    DevArr<int> dr((size_t)n_fix * n_rep, rand_idx);
    DevArr<double> dout(n_rep);
    HIPCHECK(p3d_metric_auc_borji(da.p, db.p, dr.p, n_pix, n_fix, n_rep, step_size, fidx.p, dout.p, nullptr));
    dout.get(out, n_rep);
    API_END
}
int p3d_mapf_frames(int device, const unsigned char* bgr, int n, int H0, int W0, const float mean_rgb[3], int H, int W, float* out) {
    API_BEGIN
    metric_args(device, bgr, mean_rgb, 1, 1, out);
    if (n < 1 || H0 < 1 || W0 < 1 || H < 1 || W < 1) throw P3dError("mapf: empty frame");
    DevArr<unsigned char> src((size_t)n * H0 * W0 * 3, bgr);
    DevArr<float> dst((size_t)n * H * W * 3);
    HIPCHECK(p3d_mapf_frames(src.p, n, H0, W0, dst.p, H, W, mean_rgb, nullptr));
    dst.get(out, (size_t)n * H * W * 3);
    API_END
}
int p3d_mapf_density(int device, const unsigned char* grey, int n, int H0, int W0, int H, int W, float* out) {
    API_BEGIN
    metric_args(device, grey, grey, 1, 1, out);
    if (n < 1 || H0 < 1 || W0 < 1 || H < 1 || W < 1) throw P3dError("mapf: empty frame");
    DevArr<unsigned char> src((size_t)n * H0 * W0, grey);
    DevArr<float> dst((size_t)n * H * W);
    HIPCHECK(p3d_mapf_density(src.p, n, H0, W0, dst.p, H, W, nullptr));
    dst.get(out, (size_t)n * H * W);
    API_END
}

// CRC-32C (Castagnoli) of a host buffer, slicing-by-8: the checksum of TensorFlow's checkpoint bundles
// (tensorflow/core/lib/hash/crc32c.h), used by the Python reader / writer of sap3d_tensorflow_amd/tf_checkpoint.py on the
// 248 MB of variables (train.py:180-185, 204-210, 266-267).  `crc` = running value (0 to start).
uint32_t p3d_crc32c(const void* data, size_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
        ready = true;
    }
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^
            T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int p3d_shutdown(void) {
    API_BEGIN
    if (g_live_handles.load() > 0)
        throw P3dError("p3d_shutdown with " + std::to_string(g_live_handles.load()) + " live handle(s): destroy them first (their scratch and captured graphs name the pools this call frees)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return 0;
    hipDeviceSynchronize();
    p3d_release_scratch();
    if (g_zero_page) { hipFree((void*)g_zero_page); g_zero_page = nullptr; }
    API_END
}

}  // extern "C"
