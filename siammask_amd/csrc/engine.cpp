// engine.cpp -- host side of libsiammask_hip.so: context, BN folding + weight packing,
// activation arena, the launch sequences for template / track / refine, hipGraph capture,
// and the extern "C" ABI declared in include/siammask_hip.h (product) and include/siammask_hip_test.h (tests, measurement).
//
// The network topology restated here (layer names, geometry) follows
//   experiments/siammask_sharp/resnet.py:59-103,151-227   modified ResNet-50
//   experiments/siammask_sharp/custom.py:12-25,69-159     ResDownS / UP / MaskCorr / Refine
//   models/rpn.py:41-72                                    DepthCorr
// and is cross-checked against the reference state dict by tests/test_spec.py through
// siammask_amd/spec.py (same names, same shapes).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdarg>
#include <tuple>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <vector>

#include "../../include/siammask_hip.h"
#include "../../include/siammask_hip_test.h"
#include "smk_kernels.h"

using namespace smk;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int g_concurrency_default = 0;   // measured slower on MI355X (cross-stream graph edges), see DESIGN.md

static int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(SMK_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                               \
    } while (0)

#define CHK(expr)                     \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != 0) return rc_;     \
    } while (0)

static inline int rup(int x, int a) { return (x + a - 1) / a * a; }

// 8 KB of zeros per device: every padded tap / out-of-range row / K tail of the conv kernel's
// LDS-DMA gather reads from here, which keeps the loads branch-free.
namespace smk {
const void *zero_page() {
    static void *pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        void *p = nullptr;
        if (hipMalloc(&p, 16384) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 16384) != hipSuccess) return nullptr;
        pages[dev] = p;
    }
    return pages[dev];
}
}  // namespace smk

// ---------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

// one part of a (possibly N-fused) convolution as it appears in the reference state dict
struct ConvPart {
    std::string w;      // "<name>.weight"
    std::string bn;     // BatchNorm prefix or ""
    std::string bias;   // "<name>.bias" or ""
};

struct PackedConv {
    void *w = nullptr;       // device [rows][Kpad] dtype
    void *w_halo = nullptr;  // same weights, K ordered (chunk, kh, kw, c in chunk) for conv3x3_halo_kernel (3x3 only)
    void *w_frag = nullptr;  // same weights in MFMA-fragment order for conv_wreg_kernel (f16 only)
    void *w_frag_halo = nullptr;  // 3x3, f16: the chunk-major matrix (w_halo's K order) in MFMA-fragment order (wreg_halo_tile, sequences)
    void *w_frag16 = nullptr; // small packs (<= 256 rows, K <= 640: layer1): fragment order of v_mfma_f32_16x16x32_f16 (l1_block_kernel)
    float *bias = nullptr;   // device [rows] f32
    int N = 0;               // real output channels per group
    int rows = 0;            // total rows (all groups), multiple of NPAD_ALIGN
    int group_rows = 0;      // rows per group
    int groups = 1;
    int Ci = 0, k = 1, K = 0, Kpad = 0;
    int kw = 0;              // horizontal taps when != k (pixel-pair stem)
    int alg_k = 0;           // algorithmic K (real multiply-accumulates per output) when the pack pads K
    float *oscale = nullptr; // DT_F16X3: device [rows] f32, the inverse of the power-of-two scale each row of the split pack carries (ConvParams::oscale)
    int x3_ct = 0, x3_nreal = 0;   // DT_F16X3, w_frag in FUSED order (ConvParams::x3_ct): channels / 64, activation tiles per K loop
    bool x3 = false;         // DT_F16X3: K tripled -- per tap [w_hi | w_lo | w_hi] against the operand [hi | hi | lo] gathered from the stored planes [hi | lo]; Ci = 3 x channels
};

constexpr size_t KS_PART_FLOATS = 8u << 20;      // 32 MB: e.g. 256 tiles x 4 parts x 64x128
constexpr int KS_CNT = 8192;
constexpr size_t DEC_SCRATCH_PER_STREAM = 8 * 8 + 64 * 8 + 8 * 4 + 4;     // decode_kernel's cross-workgroup scratch

// The forms of the pipelined step that measured slower than the default (a cross-queue event join, hipStreamWaitValue32, eager launches,
// the two other depth-2 forms; profiles/r05a/e/f/j/o_*) are A/B arms of measurement builds only: in the product library the four knobs are
// compile-time constants, so the arms are not even compiled.
#ifdef SMK_MEASURE
#define PIPE_JOIN (g_tune.pipe_join)
#define PIPE_SIG (g_tune.pipe_sig)
#define PIPE_EAGER (g_tune.pipe_eager)
#define PIPE_TWO_FORM (g_tune.pipe_two_form)
#else
constexpr int PIPE_JOIN = 1, PIPE_SIG = 2, PIPE_EAGER = 0, PIPE_TWO_FORM = 1;
#endif

static size_t esize(int dtype) { return dtype == DT_F32 ? 4 : 2; }
// DT_F16X3 contexts (smk_kernels.h): the KERNELS are the fp16 ones; what changes is which packs exist and how many channel planes a tensor has
static int kdtype(int dtype) { return dtype == DT_F16X3 ? DT_F16 : dtype; }

// host-side packing of ONE weight tensor [Cout][Cin][k][k] (already scaled) into rows of a
// [rows][Kpad] matrix with K ordered (ky, kx, cin_padded)
static void pack_rows(std::vector<float> &dst, int row0, int Kpad, const float *w, const double *scale,
                      int Cout, int Cin, int k, int Ci) {
    for (int n = 0; n < Cout; ++n)
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx) {
                    const double v = (double)w[(((size_t)n * Cin + ci) * k + ky) * k + kx] * scale[n];
                    dst[(size_t)(row0 + n) * Kpad + (size_t)(ky * k + kx) * Ci + ci] = (float)v;
                }
}

// fragment-order copy of an f16 pack for conv_wreg_kernel: one contiguous KB per (32 rows, 16 k) MFMA operand --
// [rows/32][Kpad/16][lane 0..63][8 halves] with lane = (n % 32) + 32 * ((k % 16) / 8), element e = k % 8
static int upload_frag_pack(PackedConv &pc, const std::vector<float> &rows_f32, int dtype) {
    if (dtype != DT_F16 || pc.rows % 32 || pc.Kpad % 16) return 0;
    const int KS16 = pc.Kpad / 16;
    std::vector<_Float16> h((size_t)pc.rows * pc.Kpad);
    for (int n = 0; n < pc.rows; ++n) {
        const float *src = rows_f32.data() + (size_t)n * pc.Kpad;
        const size_t blk = (size_t)(n / 32) * KS16;
        for (int k = 0; k < pc.Kpad; ++k) {
            const int lane = (n % 32) + 32 * ((k % 16) / 8);
            h[((blk + k / 16) * 64 + lane) * 8 + k % 8] = (_Float16)src[k];
        }
    }
    HIPCHK(hipMalloc(&pc.w_frag, h.size() * 2));
    HIPCHK(hipMemcpy(pc.w_frag, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    if (pc.rows <= 256 && pc.Kpad <= 640 && pc.Kpad % 32 == 0 && pc.groups == 1) {
        // [rows/16][Kpad/32][lane 0..63][8 halves] with lane = (n % 16) + 16 * ((k % 32) / 8), element e = k % 8
        const int KS32 = pc.Kpad / 32;
        std::vector<_Float16> g((size_t)pc.rows * pc.Kpad);
        for (int n = 0; n < pc.rows; ++n) {
            const float *src = rows_f32.data() + (size_t)n * pc.Kpad;
            const size_t blk = (size_t)(n / 16) * KS32;
            for (int k = 0; k < pc.Kpad; ++k) {
                const int lane = (n % 16) + 16 * ((k % 32) / 8);
                g[((blk + k / 32) * 64 + lane) * 8 + k % 8] = (_Float16)src[k];
            }
        }
        HIPCHK(hipMalloc(&pc.w_frag16, g.size() * 2));
        HIPCHK(hipMemcpy(pc.w_frag16, g.data(), g.size() * 2, hipMemcpyHostToDevice));
    }
    return 0;
}

static int upload_packed(PackedConv &pc, const std::vector<float> &rows_f32, const std::vector<float> &bias,
                         int dtype, const std::vector<float> *frag_rows = nullptr) {
    const size_t n = rows_f32.size();
    HIPCHK(hipMalloc(&pc.w, n * esize(dtype)));
    if (dtype == DT_F16) {
        std::vector<_Float16> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (_Float16)rows_f32[i];
        HIPCHK(hipMemcpy(pc.w, h.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMemcpy(pc.w, rows_f32.data(), n * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void **)&pc.bias, bias.size() * 4));
    HIPCHK(hipMemcpy(pc.bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    return upload_frag_pack(pc, frag_rows ? *frag_rows : rows_f32, dtype);       // derived copy for conv_wreg_kernel (f16 only)
}

// chunk-major copy of a 3x3 pack: k = (tap*Ci + c)  ->  k' = ((c / CH)*9 + tap)*CH + c % CH
static int upload_halo_pack(PackedConv &pc, const std::vector<float> &rows_f32, int dtype) {
    const int CH = dtype == DT_F16 ? 64 : 32;
    if (pc.k != 3 || pc.kw != 0 || pc.Ci % CH != 0) return 0;
    std::vector<float> hp(rows_f32.size(), 0.f);
    for (int n = 0; n < pc.rows; ++n)
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < pc.Ci; ++ci)
                hp[(size_t)n * pc.Kpad + (size_t)((ci / CH) * 9 + tap) * CH + ci % CH] =
                    rows_f32[(size_t)n * pc.Kpad + (size_t)tap * pc.Ci + ci];
    const size_t cnt = hp.size();
    if (dtype == DT_F16 && pc.rows % 32 == 0 && pc.Kpad % 16 == 0 && pc.Kpad == 9 * pc.Ci) {
        // the same matrix in fragment order (upload_frag_pack's layout) for the patch-sharing tile of the sequences
        const int KS16 = pc.Kpad / 16;
        std::vector<_Float16> h(cnt);
        for (int n = 0; n < pc.rows; ++n) {
            const float *src = hp.data() + (size_t)n * pc.Kpad;
            const size_t blk = (size_t)(n / 32) * KS16;
            for (int k = 0; k < pc.Kpad; ++k) {
                const int lane = (n % 32) + 32 * ((k % 16) / 8);
                h[((blk + k / 16) * 64 + lane) * 8 + k % 8] = (_Float16)src[k];
            }
        }
        HIPCHK(hipMalloc(&pc.w_frag_halo, cnt * 2));
        HIPCHK(hipMemcpy(pc.w_frag_halo, h.data(), cnt * 2, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc(&pc.w_halo, cnt * esize(dtype)));
    if (dtype == DT_F16) {
        std::vector<_Float16> h(cnt);
        for (size_t i = 0; i < cnt; ++i) h[i] = (_Float16)hp[i];
        HIPCHK(hipMemcpy(pc.w_halo, h.data(), cnt * 2, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMemcpy(pc.w_halo, hp.data(), cnt * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct Act {
    void *p = nullptr;
    int H = 0, W = 0, C = 0;   // C = channel stride
};

typedef std::tuple<int, int, int, std::vector<const void *>> GraphKey;

struct smk_ctx {
    int device = 0, dtype = DT_F32, variant = SMK_VARIANT_SHARP, maxB = 1;
    std::map<std::string, HostTensor> host_w;
    bool finalized = false;
    int template_B = 0;       // batch of the cached template (0 = none)
    int track_B = 0;          // batch of the last track with SMK_TRACK_MASK
    int last_B = 0, last_S = 0, last_nb = 0;
    const char *p3_buf = "a";   // arena buffer that holds the layer3 output of the last backbone run (debug read-back)
    bool graph_mode = false;
    hipStream_t cap_stream = nullptr;
    std::map<GraphKey, hipGraphExec_t> graphs;
    std::map<GraphKey, unsigned long long> graph_used;   // last use (monotonic tick): least-recently-used eviction
    unsigned long long graph_tick = 0;

    // packed convolutions
    std::map<std::string, PackedConv> conv;   // keyed by short layer id
    // arena (sized for maxB)
    std::map<std::string, void *> buf;
    std::map<std::string, size_t> buf_elems;  // per item
    std::set<std::string> buf_alias;          // names that share another entry's allocation (never freed themselves)
    int *pos_dev = nullptr;
    void *dec_scratch = nullptr;     // decode: per-stream winners of the A workgroups + arrival counters
    float *ks_part = nullptr;        // split-K: f32 partial tiles (KS_PART_FLOATS) and per-tile arrival counters
    unsigned *ks_cnt = nullptr;
    // persistent per-XCD convolution sequences (conv_seq_kernel)
    unsigned *seq_bar = nullptr;     // [8][32] u32 team counters, zero between launches
    float *seq_xch = nullptr;        // SEQ_XCH_BYTES: partial-sum slabs of the pair-split tiles (f16 contexts)
    int *seq_err = nullptr;          // device flag written by the kernel (placement / barrier timeout)
    int *seq_err_host = nullptr;     // the same flag in host-mapped pinned memory: read at every entry point without a sync
    int *seq_err_hdev = nullptr;     // device address of seq_err_host
    int seq_fail = 0;                // last failure code taken from the flag (sticky, reported by smk_seq_status)
    bool seq_pending = false;        // a sequence launch has been enqueued since the flag was last checked behind a synchronisation
    bool cap_has_seq = false;        // the graph being captured contains a sequence launch
    std::map<GraphKey, bool> graph_has_seq;
    unsigned long long *seq_clk = nullptr, *seq_clk2 = nullptr;   // SMK_SEQ_CLK stamps (measurement aid), per context
    int seq_grid = 0;                // workgroups of a sequence launch (= CUs) when the placement check passed, else 0
    bool seq_on = false;             // run_conv records into seq_rec instead of launching
    // fused frame step: the mask head is handed to the Refine chain launch (chain_mask_kernel) instead of its own launch
    bool defer_mask_req = false, have_deferred_mask = false;
    ConvParams deferred_mask;
    double deferred_mask_flop = 0.0, deferred_mask_bytes = 0.0;
    std::vector<SeqLayer> seq_rec;
    std::vector<const void *> seq_wstd;      // per record: the (kh, kw, cin)-ordered fragment pack (seq_fuse_triples needs it where the record carries the chunk-major one)
    std::vector<std::string> seq_ids;
    double seq_flop = 0.0, seq_bytes = 0.0;

    // result ring (smk_set_result_ring): caller-owned rows, library-owned cursor
    double *ring_box = nullptr;
    void *ring_ref = nullptr;
    int ring_rows = 0;
    int *ring_cursor = nullptr;      // device [4]: [0] frames committed, [1] arrival counter of the launch that advances it; [2] / [3] the same for the box
                                     // rows alone while frame steps are pipelined two deep (decode then runs ahead of the previous frame's chain)
    bool ring_in_step = false;       // smk_step is recording: decode / the Refine chain take the ring writes with them
    bool ring_step_refine = false;   // ... and a Refine launch follows the decode launch
    bool ring_ref_folded = false;    // the chain launch took the fp16 logits + the cursor

    // decode (tools/test.py:205-254 on device)
    float anchor_w[8] = {104, 88, 64, 40, 32}, anchor_h[8] = {32, 40, 64, 80, 96};   // utils/anchors.py:40-50
    int anchor_stride = 8;
    double *window_dev = nullptr;        // [25*25] outer(hanning(25), hanning(25))
    double penalty_k = 0.04, window_influence = 0.4;   // config_davis.json hp

    // fork/join concurrency between independent launches (side streams + event pool)
    bool concurrency = true;
    hipStream_t side[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;

    // per-launch profiling (smk_profile): HIP events around every kernel, eager mode only
    bool prof = false;
    bool prof_merge = false;         // smk_profile(ctx, 2): keep the merged launches of the timed path (attribution per LAUNCH, not per layer)
    struct ProfRec { std::string id, kernel; double flop, bytes; hipEvent_t e0, e1; double ext_bytes = 0.0; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_pool_next = 0;
    bool mask_join_pending = false;
    bool prof_split_l1 = false;      // (reserved) per-layer attribution of layer1 while profiling

    // software-pipelined frame steps (smk_set_pipeline): the Refine / mask tail of frame f runs on pipe_stream beside the
    // stem + layer1 launches of frame f + 1, which write the OTHER copy of p0 / p1 (the only tensors both sides touch)
    int pipe_depth = 0;              // 0 = off, 1 = one tail in flight
    int parity_now = 0;              // copy of p0 / p1 that act() resolves to (0 for every serial entry point)
    int pipe_parity = 0;             // copy the next pipelined step writes
    int last_parity = 0;             // copy the last tracked frame's p0 / p1 live in (smk_refine, smk_debug_read)
    hipStream_t pipe_stream = nullptr;
    std::vector<hipEvent_t> pipe_ev; // ring of events: "decode of frame f enqueued" / "tail of frame f enqueued"
    size_t pipe_ev_next = 0;
    bool tail_pending = false;       // a tail has been enqueued on pipe_stream and nothing has been ordered behind it yet
    hipEvent_t tail_ev = nullptr;
    int ring_batch = 0;              // batch the result ring was sized for (smk_set_result_ring)
    bool pipe_tail_has_mask = false; // (A/B knob pipe_eager bit 1) mid's capture handed the mask head to the tail
    unsigned *pipe_cnt = nullptr;    // device [16] u32: [0] semaphore "tails completed" (starts at 1), [2] semaphore "main parts completed",
                                     // [4] / [5] arrival counters of decode's streams / chain_mask's workgroups, [8] "this step's main part is running" (arms the tail gate's clock; misc_kernels.hip pipe_*)
    bool pipe_two = false;           // (recording a depth-2 pipelined step) decode keeps its own ring cursor
    bool pipe_corr_sem = false;      // ... or corr_head's first workgroup does (form 2)
    bool pipe_seq_exit = false, pipe_seq_exit_done = false;   // ... and the sequence launch raises the "chip is free" semaphore when it leaves
    bool tail2_pending = false;      // depth 2: the second part of the last frame's tail (chain + mask head) has not been launched yet
    GraphKey tail2_key, tail2_gated_key;   // ... its graph without / with the gate (flush form / the form the next step launches)
    int wave_prio_now = 0;           // (recording a pipelined step's main part) the wave priority its launches carry: smk_tune main_prio
    bool pipe_gate_late = false;     // (recording a pipelined step) seq_track launches the main gate in front of the heads
    bool pipe_mark_fold = false, pipe_tail_fold = false, pipe_done_folded = false;   // (while a pipelined step's parts are being recorded)
    unsigned *pipe_sig = nullptr;    // signal memory: main parts completed (pipe_mark_kernel); the tail's hipStreamWaitValue32 target
    unsigned pipe_sig_n = 0;         // main parts enqueued since the counter was zeroed
};

static const char *dtname(int dt) { return dt == DT_F16 ? "f16" : "f32"; }

struct ProfScope {
    smk_ctx *c; hipStream_t s; int idx = -1;
    ProfScope(smk_ctx *c_, hipStream_t s_, const std::string &id, const std::string &kernel, double flop,
              double bytes) : c(c_), s(s_) {
        if (!c->prof) return;
        smk_ctx::ProfRec r{id, kernel, flop, bytes, nullptr, nullptr, 0.0};
        // events come from a pool created by smk_profile(1): creating them between launches
        // stalled the stream at a fixed position (one layer read 10x too long)
        if (c->prof_pool_next + 2 > c->prof_pool.size()) return;
        r.e0 = c->prof_pool[c->prof_pool_next++];
        r.e1 = c->prof_pool[c->prof_pool_next++];
        (void)hipEventRecord(r.e0, s);
        c->prof_recs.push_back(r);
        idx = (int)c->prof_recs.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->prof_recs[idx].e1, s); }
    // bytes that have to cross the XCD's fabric port even when every inter-layer tensor of the launch stays in its L2
    void ext_bytes(double b) { if (idx >= 0) c->prof_recs[idx].ext_bytes = b; }
    void cancel() { if (idx >= 0 && idx == (int)c->prof_recs.size() - 1) { c->prof_recs.pop_back(); c->prof_pool_next -= 2; } idx = -1; }
};

// make `to` wait for everything enqueued so far on `from` (captured as a graph dependency)
static int stream_dep(smk_ctx *c, hipStream_t from, hipStream_t to) {
    hipEvent_t e = c->ev_pool[c->ev_next++ % c->ev_pool.size()];
    HIPCHK(hipEventRecord(e, from));
    HIPCHK(hipStreamWaitEvent(to, e, 0));
    return 0;
}
static bool parallel_ok(const smk_ctx *c) { return c->concurrency && !c->prof && c->side[0] && c->side[1]; }

static int nbranch(const smk_ctx *c) { return c->variant == SMK_VARIANT_RPN ? 2 : 3; }

static const HostTensor *find_w(const smk_ctx *c, const std::string &name) {
    auto it = c->host_w.find(name);
    return it == c->host_w.end() ? nullptr : &it->second;
}

// fold BN (eval semantics) into per-channel scale/shift, or take the conv bias
static int fold(const smk_ctx *c, const ConvPart &part, int Cout, std::vector<double> &scale,
                std::vector<double> &shift) {
    scale.assign(Cout, 1.0);
    shift.assign(Cout, 0.0);
    if (!part.bn.empty()) {
        const HostTensor *g = find_w(c, part.bn + ".weight"), *b = find_w(c, part.bn + ".bias");
        const HostTensor *m = find_w(c, part.bn + ".running_mean"), *v = find_w(c, part.bn + ".running_var");
        if (!g || !b || !m || !v) return fail(SMK_E_WEIGHT, "missing BatchNorm tensors for %s", part.bn.c_str());
        if ((int)g->data.size() != Cout || (int)b->data.size() != Cout || (int)m->data.size() != Cout ||
            (int)v->data.size() != Cout)
            return fail(SMK_E_WEIGHT, "BatchNorm %s has wrong size", part.bn.c_str());
        for (int i = 0; i < Cout; ++i) {
            const double inv = (double)g->data[i] / std::sqrt((double)v->data[i] + 1e-5);
            scale[i] = inv;
            shift[i] = (double)b->data[i] - (double)m->data[i] * inv;
        }
    }
    if (!part.bias.empty()) {
        const HostTensor *b = find_w(c, part.bias);
        if (!b || (int)b->data.size() != Cout) return fail(SMK_E_WEIGHT, "missing/mis-sized %s", part.bias.c_str());
        for (int i = 0; i < Cout; ++i) shift[i] += (double)b->data[i];
    }
    return 0;
}

// pack `parts` either N-fused (one group, rows concatenated) or as separate groups
// x3 (DT_F16X3 contexts, the layers of the track path's trunk): per tap the Ci0 = rup(Cin, 8) channels three times -- [w_hi | w_lo | w_hi]
// with w_hi = fp16(w), w_lo = fp16(w - w_hi) -- against activations stored as [hi | hi | lo] planes
// Every row is first scaled by a power of two that takes its largest |w| into [2^13, 2^14): w_lo = fp16(s w - w_hi) is then a NORMAL fp16
// number for every weight within 2^-16 of the row's maximum, i.e. the pair carries 22 bits where the unscaled, BN-folded weights (1e-2 .. 1e-3)
// leave w_lo in the subnormals (3e-6 .. 3e-5 relative).  oscale[n] = 1 / s_n (exact); the epilogue multiplies the accumulator by it.
static void split_rows_x3(std::vector<float> &rows, int nrows, int Kpad1, int Kpad3, int taps, int Ci0, std::vector<float> &oscale) {
    std::vector<float> out((size_t)nrows * Kpad3, 0.f);
    oscale.assign(nrows, 1.f);
    for (int n = 0; n < nrows; ++n) {
        float mx = 0.f;
        for (int k = 0; k < taps * Ci0; ++k) mx = std::max(mx, std::fabs(rows[(size_t)n * Kpad1 + k]));
        int e = 0;
        if (mx > 0.f) {
            e = 13 - (int)std::floor(std::log2(mx));                 // 2^e mx in [2^13, 2^14)
            e = std::max(-8, std::min(e, 40));
        }
        const float sc = std::ldexp(1.f, e);
        oscale[n] = std::ldexp(1.f, -e);
        for (int t = 0; t < taps; ++t)
            for (int ci = 0; ci < Ci0; ++ci) {
                const float v = rows[(size_t)n * Kpad1 + (size_t)t * Ci0 + ci] * sc;
                const float hi = (float)(_Float16)v, lo = (float)(_Float16)(v - hi);
                float *d = out.data() + (size_t)n * Kpad3 + (size_t)t * 3 * Ci0 + ci;
                d[0] = hi; d[Ci0] = lo; d[2 * Ci0] = hi;
            }
    }
    rows.swap(out);
}
// FUSED order of a split pack for conv_wreg_kernel (wreg_tile.inc x3ct): per tap the 3 CT tiles of 64 weights [w_hi_0 .. | w_lo_0 .. | w_hi_0 ..] become
// (w_hi_0, w_lo_0, w_hi_1, w_lo_1, .., w_hi_{CT-1}, w_lo_{CT-1}, then w_hi_0 .. w_hi_{CT-1} for the lo plane): the two products of a hi activation tile are neighbours
// in the weight stream.  Needs Ci0 % 64 == 0.  Sets pc.x3_ct / pc.x3_nreal; returns false (and leaves `out` alone) when the pack cannot be fused.
static bool fuse_rows_x3(const std::vector<float> &rows, PackedConv &pc, int taps, int Ci0, std::vector<float> &out) {
    if (Ci0 < 64 || Ci0 % 64) return false;
    const int CT = Ci0 / 64, Kp = pc.Kpad;
    out.assign(rows.size(), 0.f);
    for (int n = 0; n < pc.rows; ++n) {
        const float *s = rows.data() + (size_t)n * Kp;
        float *d = out.data() + (size_t)n * Kp;
        for (int t = 0; t < taps; ++t) {
            const float *st = s + (size_t)t * 3 * Ci0;
            float *dt = d + (size_t)t * 3 * Ci0;
            for (int j = 0; j < CT; ++j) {
                memcpy(dt + (size_t)(2 * j) * 64, st + (size_t)j * 64, 64 * sizeof(float));                       // w_hi_j   (x hi_j)
                memcpy(dt + (size_t)(2 * j + 1) * 64, st + (size_t)Ci0 + (size_t)j * 64, 64 * sizeof(float));     // w_lo_j   (x hi_j again)
                memcpy(dt + (size_t)(2 * CT + j) * 64, st + (size_t)2 * Ci0 + (size_t)j * 64, 64 * sizeof(float)); // w_hi_j   (x lo_j)
            }
        }
    }
    pc.x3_ct = CT;
    pc.x3_nreal = 0;
    for (int w = 0; w < Kp / 64; ++w) {
        const int r = w % (3 * CT);
        if (!(r < 2 * CT && (r & 1))) ++pc.x3_nreal;       // (the same cyclic rule the consumers apply, K padding included)
    }
    return true;
}
static int upload_oscale(PackedConv &pc, const std::vector<float> &oscale) {
    HIPCHK(hipMalloc((void **)&pc.oscale, oscale.size() * 4));
    HIPCHK(hipMemcpy(pc.oscale, oscale.data(), oscale.size() * 4, hipMemcpyHostToDevice));
    return 0;
}
static int pack_conv(smk_ctx *c, const std::string &id, const std::vector<ConvPart> &parts, int Cin, int Cout,
                     int k, bool grouped, bool x3 = false) {
    PackedConv pc;
    pc.Ci = rup(Cin, 8);
    pc.k = k;
    pc.K = k * k * pc.Ci;
    pc.Kpad = rup(pc.K, KPAD_ALIGN);
    const int np = (int)parts.size();
    if (grouped) {
        pc.groups = np;
        pc.N = Cout;
        pc.group_rows = rup(Cout, NPAD_ALIGN);
        pc.rows = pc.group_rows * np;
    } else {
        pc.groups = 1;
        pc.N = Cout * np;
        pc.group_rows = pc.rows = rup(Cout * np, NPAD_ALIGN);
    }
    std::vector<float> rows((size_t)pc.rows * pc.Kpad, 0.f), bias(pc.rows, 0.f);
    for (int i = 0; i < np; ++i) {
        const HostTensor *w = find_w(c, parts[i].w);
        if (!w) return fail(SMK_E_WEIGHT, "missing weight %s", parts[i].w.c_str());
        if (w->shape.size() != 4 || w->shape[0] != Cout || w->shape[1] != Cin || w->shape[2] != k || w->shape[3] != k)
            return fail(SMK_E_WEIGHT, "weight %s has wrong shape", parts[i].w.c_str());
        std::vector<double> scale, shift;
        CHK(fold(c, parts[i], Cout, scale, shift));
        const int row0 = grouped ? i * pc.group_rows : i * Cout;
        pack_rows(rows, row0, pc.Kpad, w->data.data(), scale.data(), Cout, Cin, k, pc.Ci);
        for (int n = 0; n < Cout; ++n) bias[row0 + n] = (float)shift[n];
    }
    if (x3) {
        const int Ci0 = pc.Ci, K1 = pc.Kpad;
        pc.x3 = true;
        pc.alg_k = k * k * Ci0;
        pc.Ci = 3 * Ci0;
        pc.K = k * k * pc.Ci;
        pc.Kpad = rup(pc.K, KPAD_ALIGN);
        std::vector<float> osc;
        split_rows_x3(rows, pc.rows, K1, pc.Kpad, k * k, Ci0, osc);
        CHK(upload_oscale(pc, osc));
        std::vector<float> fused;
        if (g_tune.x3_fused && fuse_rows_x3(rows, pc, k * k, Ci0, fused)) {
            CHK(upload_packed(pc, rows, bias, kdtype(c->dtype), &fused));
            c->conv[id] = pc;
            return 0;
        }
    }
    CHK(upload_packed(pc, rows, bias, kdtype(c->dtype)));
    if (!grouped && !x3) CHK(upload_halo_pack(pc, rows, kdtype(c->dtype)));
    c->conv[id] = pc;
    return 0;
}

static ConvPart bnpart(const std::string &conv, const std::string &bn) { return ConvPart{conv + ".weight", bn, ""}; }
static ConvPart biaspart(const std::string &conv) { return ConvPart{conv + ".weight", "", conv + ".bias"}; }

// refine_model.deconv: ConvTranspose2d(256, 32, 15, 15) on a 1x1 input == GEMM with
// N = 15*15*32 ordered (ky, kx, co) so that the result is the NHWC tensor [15][15][32]
static int pack_deconv(smk_ctx *c) {
    const HostTensor *w = find_w(c, "refine_model.deconv.weight"), *b = find_w(c, "refine_model.deconv.bias");
    if (!w || !b) return fail(SMK_E_WEIGHT, "missing refine_model.deconv.*");
    if (w->shape.size() != 4 || w->shape[0] != 256 || w->shape[1] != 32 || w->shape[2] != 15 || w->shape[3] != 15)
        return fail(SMK_E_WEIGHT, "refine_model.deconv.weight has wrong shape");
    PackedConv pc;
    pc.Ci = 256; pc.k = 1; pc.K = 256; pc.Kpad = 256; pc.groups = 1;
    pc.N = 15 * 15 * 32;
    pc.group_rows = pc.rows = rup(pc.N, NPAD_ALIGN);
    std::vector<float> rows((size_t)pc.rows * pc.Kpad, 0.f), bias(pc.rows, 0.f);
    for (int ci = 0; ci < 256; ++ci)
        for (int co = 0; co < 32; ++co)
            for (int p = 0; p < 225; ++p) {
                const int n = p * 32 + co;
                rows[(size_t)n * pc.Kpad + ci] = w->data[((size_t)ci * 32 + co) * 225 + p];
            }
    for (int p = 0; p < 225; ++p)
        for (int co = 0; co < 32; ++co) bias[p * 32 + co] = b->data[co];
    CHK(upload_packed(pc, rows, bias, kdtype(c->dtype)));
    c->conv["deconv"] = pc;
    return 0;
}

// features.conv1 (7x7 s2 p0, 3 -> 64) on the pixel-pair input layout [H][ceil(W/2)][2 px x 4 ch]:
// a 7 x 4 convolution over pixel pairs (vertical stride 2, horizontal stride 1 pair), K = 7*4*8 = 224
// instead of 7*7*8 = 392 with the channel-padded layout; the 8th pixel and the 4th channel have zero weights
static int pack_stem(smk_ctx *c) {
    const std::string f = "features.features.";
    const HostTensor *w = find_w(c, f + "conv1.weight");
    if (!w) return fail(SMK_E_WEIGHT, "missing weight %sconv1.weight", f.c_str());
    if (w->shape.size() != 4 || w->shape[0] != 64 || w->shape[1] != 3 || w->shape[2] != 7 || w->shape[3] != 7)
        return fail(SMK_E_WEIGHT, "weight %sconv1.weight has wrong shape", f.c_str());
    std::vector<double> scale, shift;
    CHK(fold(c, bnpart(f + "conv1", f + "bn1"), 64, scale, shift));
    PackedConv pc;
    pc.Ci = 8; pc.k = 7; pc.kw = 4; pc.K = 7 * 4 * 8; pc.Kpad = rup(pc.K, KPAD_ALIGN); pc.alg_k = 3 * 7 * 7;
    pc.groups = 1; pc.N = 64; pc.group_rows = pc.rows = rup(64, NPAD_ALIGN);
    std::vector<float> rows((size_t)pc.rows * pc.Kpad, 0.f), bias(pc.rows, 0.f);
    for (int n = 0; n < 64; ++n) {
        for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx)
                for (int ci = 0; ci < 3; ++ci) {
                    const double v = (double)w->data[(((size_t)n * 3 + ci) * 7 + ky) * 7 + kx] * scale[n];
                    rows[(size_t)n * pc.Kpad + (size_t)(ky * 4 + kx / 2) * 8 + (kx & 1) * 4 + ci] = (float)v;
                }
        bias[n] = (float)shift[n];
    }
    CHK(upload_packed(pc, rows, bias, kdtype(c->dtype)));
    c->conv["stem"] = pc;
    return 0;
}

static const int STAGE_PLANES[3] = {64, 128, 256};
static const int STAGE_BLOCKS[3] = {3, 4, 6};


static int build_weights(smk_ctx *c) {
    const std::string f = "features.features.";
    const bool x3 = c->dtype == DT_F16X3;              // the trunk of the track path in split operands; mask head + Refine stay plain fp16
    if (x3) CHK(pack_conv(c, "stem", {bnpart(f + "conv1", f + "bn1")}, 3, 64, 7, false, true));      // (generic 7x7 on the operand [hi | hi | lo] x 8 channels)
    else CHK(pack_stem(c));
    int inplanes = 64;
    for (int s = 0; s < 3; ++s) {
        const int planes = STAGE_PLANES[s];
        for (int b = 0; b < STAGE_BLOCKS[s]; ++b) {
            char pre[64], id[32];
            snprintf(pre, sizeof(pre), "%slayer%d.%d.", f.c_str(), s + 1, b);
            snprintf(id, sizeof(id), "l%d.%d.", s + 1, b);
            const std::string p = pre, i = id;
            const int cin = b == 0 ? inplanes : planes * 4;
            CHK(pack_conv(c, i + "c1", {bnpart(p + "conv1", p + "bn1")}, cin, planes, 1, false, x3));
            CHK(pack_conv(c, i + "c2", {bnpart(p + "conv2", p + "bn2")}, planes, planes, 3, false, x3));
            CHK(pack_conv(c, i + "c3", {bnpart(p + "conv3", p + "bn3")}, planes, planes * 4, 1, false, x3));
            if (b == 0)
                CHK(pack_conv(c, i + "ds", {bnpart(p + "downsample.0", p + "downsample.1")}, cin, planes * 4,
                              s == 0 ? 1 : 3, false, x3));
        }
        inplanes = planes * 4;
    }
    CHK(pack_conv(c, "adjust", {bnpart("features.downsample.downsample.0", "features.downsample.downsample.1")},
                  1024, 256, 1, false, x3));
    std::vector<std::string> br = {"rpn_model.cls.", "rpn_model.loc."};
    if (c->variant != SMK_VARIANT_RPN) br.push_back("mask_model.mask.");
    std::vector<ConvPart> ck, cs, h0;
    for (auto &b : br) {
        ck.push_back(bnpart(b + "conv_kernel.0", b + "conv_kernel.1"));
        cs.push_back(bnpart(b + "conv_search.0", b + "conv_search.1"));
        h0.push_back(bnpart(b + "head.0", b + "head.1"));
    }
    CHK(pack_conv(c, "conv_kernel", ck, 256, 256, 3, false, x3));   // N-fused: [cls | loc | mask]
    CHK(pack_conv(c, "conv_search", cs, 256, 256, 3, false, x3));
    CHK(pack_conv(c, "head0", h0, 256, 256, 1, true, x3));          // grouped: each branch its own input
    CHK(pack_conv(c, "cls3", {biaspart("rpn_model.cls.head.3")}, 256, 10, 1, false, x3));
    CHK(pack_conv(c, "loc3", {biaspart("rpn_model.loc.head.3")}, 256, 20, 1, false, x3));
    if (c->variant != SMK_VARIANT_RPN)
        CHK(pack_conv(c, "mask3", {biaspart("mask_model.mask.head.3")}, 256, 63 * 63, 1, false));
    if (c->variant == SMK_VARIANT_SHARP) {
        struct R { const char *n; int c0, c1, c2; };
        const R rs[6] = {{"v0", 64, 16, 4}, {"v1", 256, 64, 16}, {"v2", 512, 128, 32},
                         {"h2", 32, 32, 32}, {"h1", 16, 16, 16}, {"h0", 4, 4, 4}};
        for (auto &r : rs) {
            const std::string p = std::string("refine_model.") + r.n;
            CHK(pack_conv(c, std::string(r.n) + ".0", {biaspart(p + ".0")}, r.c0, r.c1, 3, false));
            CHK(pack_conv(c, std::string(r.n) + ".2", {biaspart(p + ".2")}, r.c1, r.c2, 3, false));
        }
        CHK(pack_conv(c, "post0", {biaspart("refine_model.post0")}, 32, 16, 3, false));
        CHK(pack_conv(c, "post1", {biaspart("refine_model.post1")}, 16, 4, 3, false));
        CHK(pack_conv(c, "post2", {biaspart("refine_model.post2")}, 4, 1, 3, false));
        CHK(pack_deconv(c));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// arena
// ---------------------------------------------------------------------------------------------
static int alloc_buf(smk_ctx *c, const char *name, size_t elems_per_item) {
    void *p = nullptr;
    // (DT_F16X3: two channel planes [hi | lo] per tensor of the trunk; the Refine buffers get them too -- small, and one rule)
    const size_t bytes = elems_per_item * (c->dtype == DT_F16X3 ? X3_PLANES : 1) * (size_t)c->maxB * esize(c->dtype) + 256;
    HIPCHK(hipMalloc(&p, bytes));
    HIPCHK(hipMemset(p, 0, bytes));
    c->buf[name] = p;
    c->buf_elems[name] = elems_per_item;
    return 0;
}

static int build_arena(smk_ctx *c) {
    const int nb = nbranch(c);
    CHK(alloc_buf(c, "xin", 255 * 255 * 8));
    CHK(alloc_buf(c, "p0", 125 * 125 * 64));
    CHK(alloc_buf(c, "x1", 63 * 63 * 64));
    CHK(alloc_buf(c, "t1", 63 * 63 * 128));
    CHK(alloc_buf(c, "t2", 63 * 63 * 128));
    CHK(alloc_buf(c, "r", 63 * 63 * 256));
    CHK(alloc_buf(c, "a", 63 * 63 * 256));
    CHK(alloc_buf(c, "b", 63 * 63 * 256));
    CHK(alloc_buf(c, "p1", 63 * 63 * 256));
    // layer2 / layer3 have their OWN intermediates, one layout per buffer (run_backbone): inside the persistent sequence the eight
    // teams are not synchronised with each other, and a buffer that changes its image pitch between layers lets a team that is ahead
    // write over the images of a team that is behind (found at B = 9..14: profiles/r03h_b12_race.txt).  Only f16 contexts can run
    // the sequence kernel; an f32 context launches layer after layer on one stream, where re-using a / b / t1 / t2 / r is safe, so
    // there the stage-private names are ALIASES of the shared buffers (5.7 M elements per image saved: 1.45 GB at max_batch 64).
    static const char *STAGE_PRIV[][2] = {{"t1_s", "t1"}, {"t1_2", "t1"}, {"t2_2", "t2"}, {"r_2", "r"}, {"a_2", "a"}, {"b_2", "b"},
                                          {"t1_3", "t1"}, {"t2_3", "t2"}, {"r_3", "r"}, {"a_3", "a"}, {"b_3", "b"}};
    static const size_t STAGE_PRIV_ELEMS[] = {63 * 63 * 128, 31 * 31 * 128, 31 * 31 * 128, 31 * 31 * 512, 31 * 31 * 512, 31 * 31 * 512,
                                              31 * 31 * 256, 31 * 31 * 256, 31 * 31 * 1024, 31 * 31 * 1024, 31 * 31 * 1024};
    for (size_t i = 0; i < sizeof(STAGE_PRIV) / sizeof(STAGE_PRIV[0]); ++i) {
        if (c->dtype == DT_F16) CHK(alloc_buf(c, STAGE_PRIV[i][0], STAGE_PRIV_ELEMS[i]));
        else {
            if (STAGE_PRIV_ELEMS[i] > c->buf_elems.at(STAGE_PRIV[i][1])) return fail(SMK_E_STATE, "arena alias %s", STAGE_PRIV[i][0]);
            c->buf[STAGE_PRIV[i][0]] = c->buf.at(STAGE_PRIV[i][1]);
            c->buf_elems[STAGE_PRIV[i][0]] = STAGE_PRIV_ELEMS[i];
            c->buf_alias.insert(STAGE_PRIV[i][0]);
        }
    }
    CHK(alloc_buf(c, "p2", 31 * 31 * 512));
    CHK(alloc_buf(c, "search", 31 * 31 * 256));
    CHK(alloc_buf(c, "zf", 7 * 7 * 256));
    CHK(alloc_buf(c, "zk", 5 * 5 * 256 * nb));
    CHK(alloc_buf(c, "xs", 29 * 29 * 256 * nb));
    CHK(alloc_buf(c, "corr", 25 * 25 * 256 * nb));
    CHK(alloc_buf(c, "head0", 25 * 25 * 256 * nb));
    if (c->variant == SMK_VARIANT_SHARP) {
        CHK(alloc_buf(c, "rf_d", 15 * 15 * 32));
        CHK(alloc_buf(c, "rf_h2a", 15 * 15 * 32));
        CHK(alloc_buf(c, "rf_h2b", 15 * 15 * 32));
        CHK(alloc_buf(c, "rf_v2a", 15 * 15 * 128));
        CHK(alloc_buf(c, "rf_s2", 15 * 15 * 32));
        CHK(alloc_buf(c, "rf_u0", 31 * 31 * 16));
        CHK(alloc_buf(c, "rf_h1a", 31 * 31 * 16));
        CHK(alloc_buf(c, "rf_h1b", 31 * 31 * 16));
        CHK(alloc_buf(c, "rf_v1a", 31 * 31 * 64));
        CHK(alloc_buf(c, "rf_s1", 31 * 31 * 16));
        CHK(alloc_buf(c, "rf_u1", 61 * 61 * 8));
        CHK(alloc_buf(c, "rf_h0a", 61 * 61 * 8));
        CHK(alloc_buf(c, "rf_h0b", 61 * 61 * 8));
        CHK(alloc_buf(c, "rf_v0a", 61 * 61 * 16));
        CHK(alloc_buf(c, "rf_s0", 61 * 61 * 8));
    }
    HIPCHK(hipMalloc((void **)&c->pos_dev, sizeof(int) * 2 * c->maxB));
    HIPCHK(hipMemset(c->pos_dev, 0, sizeof(int) * 2 * c->maxB));
    HIPCHK(hipMalloc((void **)&c->ks_part, KS_PART_FLOATS * sizeof(float)));
    HIPCHK(hipMalloc((void **)&c->ks_cnt, KS_CNT * sizeof(unsigned)));
    HIPCHK(hipMalloc((void **)&c->seq_bar, 8 * 32 * sizeof(unsigned)));
    HIPCHK(hipMemset(c->seq_bar, 0, 8 * 32 * sizeof(unsigned)));
    HIPCHK(hipMalloc((void **)&c->seq_err, sizeof(int)));
    HIPCHK(hipMemset(c->seq_err, 0, sizeof(int)));
    if (c->dtype == DT_F16) {
        HIPCHK(hipMalloc((void **)&c->seq_xch, SEQ_XCH_BYTES));
        HIPCHK(hipMemset(c->seq_xch, 0, SEQ_XCH_BYTES));
    }
    HIPCHK(hipHostMalloc((void **)&c->seq_err_host, 64, hipHostMallocMapped));
    *c->seq_err_host = 0;
    HIPCHK(hipHostGetDevicePointer((void **)&c->seq_err_hdev, c->seq_err_host, 0));
    HIPCHK(hipMemset(c->ks_cnt, 0, KS_CNT * sizeof(unsigned)));
    HIPCHK(hipMalloc(&c->dec_scratch, (size_t)c->maxB * DEC_SCRATCH_PER_STREAM));
    HIPCHK(hipMemset(c->dec_scratch, 0, (size_t)c->maxB * DEC_SCRATCH_PER_STREAM));
    return 0;
}

static Act act(smk_ctx *c, const char *name, int H, int W, int C) {
    Act a;
    // p0 / p1 exist twice while frame steps are pipelined (the tail of frame f reads one copy, the front of f + 1 writes the other)
    if (c->parity_now && name[0] == 'p' && (name[1] == '0' || name[1] == '1' || name[1] == '2') && !name[2])
        a.p = c->buf.at(name[1] == '0' ? "p0#1" : (name[1] == '1' ? "p1#1" : "p2#1"));
    else if (c->parity_now && c->pipe_depth >= 2 && !strcmp(name, "head0")) a.p = c->buf.at("head0#1");   // (depth 2: the mask head of frame f runs beside corr_head of f + 1)
    else
    a.p = c->buf.at(name);
    a.H = H; a.W = W; a.C = C;
    return a;
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
struct ConvOpt {
    int stride = 1, pad = 0, dil = 1, relu = 0;
    int stride_x = 0;         // horizontal stride when != stride (pixel-pair stem)
    const Act *res = nullptr;
    int res_mode = RES_NONE;
    int res_coff = 0;
    int cin_off = 0;          // channel slice of the input
    int cout_off = 0;
    int n_override = 0;       // use only the first n rows of a fused pack
    int groups = 1;
    // window / upsample view of the input
    bool win = false;
    int Hl = 0, Wl = 0, org_y = 0, org_x = 0;
    const int *pos = nullptr;
    int pos_mul = 0, pos_add = 0;
    bool ups = false;
    // NCHW f32 output
    float *nchw_out = nullptr;
    int algo_naive = 0;
    int tile_code = 0;
    int halo = 0;             // 128 / 64: force the halo kernel with this BM (per-op tests)
    int wreg = 0;             // 1..6: force conv_wreg_kernel with this tile code (per-op tests, micro-benchmark)
};

static int conv_params(const smk_ctx *c, const PackedConv &pc, const Act &in, const Act *out, int B,
                       const ConvOpt &o, ConvParams &p) {
    memset(&p, 0, sizeof(p));
    p.in = in.p;
    p.wgt = pc.w;
    p.wgt_frag = pc.w_frag;
    p.wgt_frag_halo = pc.w_frag_halo;
    p.bias = pc.bias;
    p.oscale = pc.oscale;
    p.pos = o.pos;
    p.B = B;
    p.Hs = in.H; p.Ws = in.W; p.Cs = in.C;
    p.cin_off = o.cin_off;
    p.Ci = pc.Ci;
    p.x3_ct = pc.x3_ct; p.x3_nreal = pc.x3_nreal;
    p.x3_in = pc.x3 ? pc.Ci / 3 : 0;      // split tensor in: the operand's [hi | hi | lo] channels of a tap are gathered from the stored [hi | lo] planes
    p.Hl = (o.win || o.ups) ? o.Hl : in.H;
    p.Wl = (o.win || o.ups) ? o.Wl : in.W;
    p.org_y = o.org_y; p.org_x = o.org_x;
    p.pos_mul = o.pos_mul; p.pos_add = o.pos_add;
    p.ups = o.ups ? 1 : 0;
    p.kh = pc.k;
    p.kw = pc.kw ? pc.kw : pc.k;
    p.stride = o.stride; p.pad = o.pad; p.dil = o.dil;
    p.stride_x = o.stride_x ? o.stride_x : o.stride;
    p.Ho = (p.Hl + 2 * o.pad - o.dil * (p.kh - 1) - 1) / p.stride + 1;
    p.Wo = (p.Wl + 2 * o.pad - o.dil * (p.kw - 1) - 1) / p.stride_x + 1;
    p.K = pc.K; p.Kpad = pc.Kpad;
    p.N = o.n_override ? o.n_override : pc.N;
    p.M = B * p.Ho * p.Wo;
    p.relu = o.relu;
    p.groups = o.groups;
    if (o.groups > 1) {
        p.g_cin_off = pc.x3 ? pc.Ci / 3 * X3_PLANES : pc.Ci;      // branches sit side by side in the channel dimension (split tensors: a branch's
        p.g_wgt_off = pc.group_rows;                               //  two stored planes side by side; pc.Ci is the OPERAND's channel count, 3 per value)
        p.g_cout_off = pc.group_rows * (pc.x3 ? X3_PLANES : 1);
    }
    if (o.nchw_out) {
        p.out = o.nchw_out;
        p.out_mode = OUT_NCHW_F32;
        p.Nst = rup(p.N, 4);
    } else {
        if (!out) return fail(SMK_E_ARG, "conv without output");
        if (out->H != p.Ho || out->W != p.Wo)
            return fail(SMK_E_ARG, "internal: conv output %dx%d != buffer %dx%d", p.Ho, p.Wo, out->H, out->W);
        p.out = out->p;
        p.out_mode = OUT_NHWC;
        p.Cos = out->C;
        p.cout_off = o.cout_off;
        p.Nst = rup(p.N, 8);
        if (pc.x3) {
            // split output: whole-tensor planes (stride = half of the buffer's channels), per-branch planes for a grouped launch
            p.x3_out = o.groups > 1 ? pc.group_rows : out->C / X3_PLANES;
            if (out->C % X3_PLANES || o.cout_off + (o.groups - 1) * X3_PLANES * pc.group_rows + p.x3_out + p.Nst > out->C)
                return fail(SMK_E_ARG, "internal: split conv output channels exceed buffer");
        } else if (o.cout_off + (o.groups - 1) * pc.group_rows + p.Nst > out->C)
            return fail(SMK_E_ARG, "internal: conv output channels exceed buffer");
    }
    p.xcd_mode = g_tune.xcd_mode;
    // NCHW f32 epilogue: a tile writes 128 consecutive positions of each channel plane (512 B at a 4-byte aligned, not
    // line aligned, offset), so the first / last line of every segment is completed by the NEIGHBOURING M tile.  tn-major
    // order runs the M neighbours of one channel block back to back on one XCD: the partial lines meet in that L2
    // before they are evicted (tm-major: the neighbour comes tilesN tiles later -> read-modify-write at the memory side)
    // (measured, profiles/r02_tail_ab.txt: B=64 -1.3 % on the step, B=8 / B=1 within noise -> large launches only)
    if (o.nchw_out && g_tune.nchw_tn_major && p.xcd_mode == 1 && p.M >= 20000 && (double)p.M * p.N * 4 >= 4.0e6) p.xcd_mode = 2;
    p.prio = g_tune.prio;
    p.wave_prio = c->wave_prio_now;
    {
        const double ib = (double)B * in.H * in.W * in.C * esize(c->dtype), wb = (double)pc.rows * pc.Kpad * esize(c->dtype);
        p.buf_lds = (g_tune.buf_lds && ib < 2.0e9 && wb < 2.0e9) ? 1 : 0;
        p.a_stage = g_tune.a_stage;
        p.res_nt = g_tune.res_nt;
        p.in_bytes = (unsigned)(ib < 4.0e9 ? ib : 0);
        p.w_bytes = (unsigned)(wb < 4.0e9 ? wb : 0);
    }
    p.nt_store = (g_tune.nt_store && o.nchw_out && (double)p.M * p.N * 4 >= 4.0e6) ? 1 : 0;
    p.ci_shift = -1;
    for (int sh = 0; sh < 16; ++sh)
        if ((1 << sh) == p.Ci) p.ci_shift = sh;
    p.kw_magic = 65536 / p.kw + 1;                        // exact for tap < 4096 and kw <= 15
    p.zero = c->device >= 0 ? zero_page() : nullptr;      // device < 0: host-side walk (CPU tests)
    if (c->device >= 0 && !p.zero) return fail(SMK_E_HIP, "could not allocate the zero page");
    if (o.res) {
        p.res = o.res->p;
        p.res_Cs = o.res->C;
        p.res_coff = o.res_coff;
        p.res_mode = o.res_mode;
        if (pc.x3) p.x3_res = o.res->C / X3_PLANES;
    }
    p.ksplit = 1;
    p.ks_part = c->ks_part;
    p.ks_cnt = c->ks_cnt;
    p.ks_part_cap = c->ks_part ? KS_PART_FLOATS : 0;
    p.ks_cnt_cap = c->ks_cnt ? KS_CNT : 0;
    return 0;
}

// code: bits 0-3 tile (0 auto, 1 128x128, 2 128x64, 3 64x128, 4 64x64), bits 4-5 K tile
// (0 auto, 1 128 B, 2 256 B), bits 6-7 ring depth (0 auto, 1..3 -> 2..4 stages)
static TileChoice tile_from_code(int code, const ConvParams &p, int dtype) {
    TileChoice t = choose_tile(p, dtype);
    static const int tb[6][2] = {{0, 0}, {128, 128}, {128, 64}, {64, 128}, {64, 64}, {256, 128}};
    const int tile = code & 15, kt = (code >> 4) & 3, st = (code >> 6) & 3;
    if (tile >= 1 && tile <= 5) {
        t.bm = tb[tile][0]; t.bn = tb[tile][1];
        t.kt = (t.bm == 64 && t.bn == 64) ? 256 : 128;
        t.stages = (t.bm == 128 && t.bn == 128) ? 2 : 3;
    }
    if (kt) t.kt = kt == 2 ? 256 : 128;
    if (t.bm == 64 && t.bn == 64) t.kt = 256;
    if (t.bm == 256) t.kt = 128;
    if (st) t.stages = st + 1;
    return t;
}

// conv3x3_halo_kernel or the generic kernel?  Returns the halo workgroup height (128 / 64) or 0.
// Measured on MI355X (profiles/r01_v6_halo_ab.txt): the halo kernel wins on every 3x3 stride-1 layer of the
// path except the long-K wide-N projection (l3.0.downsample, K=4608 N=1024), where the 256x128 generic tile
// amortises the weight stream better; BM=128 once the launch has >= 300 such tiles, else BM=64.
static int halo_choice(const PackedConv &pc, const ConvParams &p, const ConvOpt &o, int dtype) {
    const int mode = o.halo ? o.halo : g_tune.halo;
    if (!mode || !pc.w_halo || p.out_mode != OUT_NHWC || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.ups) return 0;
    if (mode != 1) return mode;
    if (dtype != DT_F16) return 0;            // fp32 (32-channel chunks, 32x32x2 MFMA): measured slower, 1.33 vs 1.09 ms at B=1
    if (p.Ci * 9 > 2304 && p.Nst >= 512) return 0;
    const long tiles128 = (long)p.B * ((p.Ho * p.Wo + 127) / 128) * ((p.Nst + 127) / 128);
    return tiles128 >= 300 ? 128 : 64;
}

// ---- persistent per-XCD convolution sequences (conv_seq_kernel) ---------------------------------------------------
// While c->seq_on, run_conv / run_conv_jobs RECORD eligible convolutions instead of launching them; seq_flush turns the
// recorded list into persistent launches of <= SEQ_MAX layers.  Everything else (non-eligible convolutions, other
// kernels) flushes first, so program order is preserved.
// the patch-sharing tile of the sequences (wreg_halo_tile.inc): can this 3x3 convolution run on whole-row tiles of bm pixels?
static bool seq_halo_ok(const ConvParams &p, int bm) {
    if (!p.wgt_frag_halo || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.stride_x != 1 || p.pad != p.dil || p.dil < 1 || p.dil > 4) return false;
    if (p.Hl != p.Hs || p.Wl != p.Ws || p.org_y || p.org_x || p.Ho != p.Hl || p.Wo != p.Wl) return false;
    if (p.Ci % 128 || p.Kpad != 9 * p.Ci || p.Wo > bm) return false;       // an even number of 64-channel chunks
    const int rpt = bm / p.Wo;
    return (rpt + 2 * p.dil) * (p.Wl + 2 * p.dil) * 9 <= 10 * 256;         // patch pieces (HALO_NRMAX rounds of 256; 144-byte rows)
}
// force_halo: 0 = the rule below, 128 / 64 = that tile or fail, -1 = never (per-op tests that force another tile)
static bool seq_layer_from(const ConvParams &p, int dtype, SeqLayer &L, int force_halo = 0) {
    if (!conv_wreg_eligible(p, dtype) || p.groups > 1 || p.pos || p.ups || p.Kpad % 128) return false;
    if (p.ci_shift < 0 || p.Ci < 64) return false;       // (wreg_tile compiles the general tap arithmetic out of the sequence's routines)
    if (p.kh > 15 || p.kw > 15 || p.stride > 15 || p.pad > 15 || p.dil > 15) return false;
    // the packed record keeps the geometry in 16-bit fields
    const int u16[] = {p.Hs, p.Ws, p.Cs, p.cin_off, p.Ci, p.Hl, p.Wl, p.Ho, p.Wo, p.Kpad, p.Nst, p.Cos, p.cout_off, p.res_Cs, p.res_coff};
    for (int v : u16)
        if (v < 0 || v > 65535) return false;
    if (p.org_y < -32768 || p.org_y > 32767 || p.org_x < -32768 || p.org_x > 32767) return false;
    memset(&L, 0, sizeof(L));
    L.in = p.in; L.wgt_frag = p.wgt_frag; L.bias = p.bias; L.res = p.res; L.out = p.out;
    L.in_bytes = p.in_bytes; L.w_bytes = p.w_bytes;
    L.Hs = (unsigned short)p.Hs; L.Ws = (unsigned short)p.Ws; L.Cs = (unsigned short)p.Cs; L.cin_off = (unsigned short)p.cin_off;
    L.Ci = (unsigned short)p.Ci; L.Hl = (unsigned short)p.Hl; L.Wl = (unsigned short)p.Wl;
    L.org_y = (short)p.org_y; L.org_x = (short)p.org_x; L.Ho = (unsigned short)p.Ho; L.Wo = (unsigned short)p.Wo;
    L.Kpad = (unsigned short)p.Kpad; L.Nst = (unsigned short)p.Nst; L.Cos = (unsigned short)p.Cos;
    L.cout_off = (unsigned short)p.cout_off; L.res_Cs = (unsigned short)p.res_Cs; L.res_coff = (unsigned short)p.res_coff;
    L.kw_magic = p.kw_magic;
    L.kh = (signed char)p.kh; L.kw = (signed char)p.kw; L.stride = (signed char)p.stride; L.stride_x = (signed char)p.stride_x;
    L.pad = (signed char)p.pad; L.dil = (signed char)p.dil; L.relu = (signed char)p.relu; L.res_mode = (signed char)p.res_mode;
    L.ci_shift = (signed char)p.ci_shift;
    L.a_stage = (signed char)p.a_stage;
    L.res_nt = (signed char)p.res_nt;
    // workgroup tile: the widest that still gives the 32 workgroups of an XCD a tile each per image
    L.cfg = p.Nst >= 512 ? 0 : (p.Nst >= 192 ? 1 : 2);
    // Short-K layers are dominated by the fixed cost of a tile (operand first touch, residual fetch, accumulator hand-over:
    // ~5 us against ~0.45 us per K tile, SMK_SEQ_CLK), so when the 64-row tiling needs more than one round of the team's
    // 32 workgroups per image, ONE 128-row tile per workgroup beats two 64-row tiles in sequence
    // (bottleneck conv3: 2 x 64x256 -> 1 x 128x256; layer2.0 conv1 on the 63x63 input: 4 rounds of 64x64 -> 1 of 128x128)
    // (seq_tall = 2, A/B knob: also for long-K layers -- l3.0.downsample, 64 tiles of 64x256 = two rounds -- now that four
    //  producer waves feed a 128-row tile)
    if (g_tune.seq_tall && ((long)p.kh * p.kw * p.Ci <= 512 || g_tune.seq_tall == 2)) {
        const int hw = p.Ho * p.Wo;
        const int bn64 = L.cfg == 0 ? 256 : (L.cfg == 1 ? 128 : 64);
        const int tiles64 = ((hw + 63) / 64) * ((p.Nst + bn64 - 1) / bn64);
        if (tiles64 > 32) {
            if (p.Nst >= 512) L.cfg = 3;
            else if (p.Nst >= 96) L.cfg = 4;
            else L.cfg = 9;                              // 128x64 (layer1's 64-channel convolutions on 63x63 images)
        }
        if (tiles64 > 64 && p.Nst >= 192 && p.Nst < 512) L.cfg = 3;      // N = 256 on 63x63 images: 128x256, one round (layer1 conv3)
    }
    // N = 512 with 16 row tiles (layer2.0's 3x3 stride-2 shortcut on a 31x31 output): 32 tiles either as 64x256 or as 128x128 --
    // the square tile stages 32 KB per K tile instead of 40 KB for the same flops, and these loops run at the CU's 64 B/clk
    // (smk_tune "seq_ds128", A/B knob)
    if (g_tune.seq_ds128 && L.cfg == 0 && p.Nst == 512 && (long)p.kh * p.kw * p.Ci >= 1024) {
        const int hw = p.Ho * p.Wo;
        if (((hw + 127) / 128) * 4 <= 32) L.cfg = 4;
    }
    // 3x3 stride-1 layers with N <= 256 (every Bottleneck's conv2): whole-row tiles x 64 channels with the activation patch shared
    // by the nine taps -- half the bytes per flop of the 64 x 128 / 64 x 64 im2col tiles (smk_tune "seq_halo").  128 pixels where that
    // gives the team (nearly) a tile per workgroup (256 channels on 31 x 31: 8 x 4), else 64 (128 channels: 16 x 2).  The long-K
    // wide-N shortcut of layer3.0 stays on 128 x 256 tiles (same bytes per flop, four times fewer tiles).
    if (force_halo > 0 || (force_halo == 0 && g_tune.seq_halo && p.Nst <= 256)) {
        const int tn = (p.Nst + 63) / 64;
        int bm = force_halo > 0 ? force_halo : 0;
        if (!bm) {
            const bool ok128 = seq_halo_ok(p, 128), ok64 = seq_halo_ok(p, 64);
            const int t128 = ok128 ? ((p.Ho + 128 / p.Wo - 1) / (128 / p.Wo)) * tn : 0;
            bm = (ok128 && (t128 >= 28 || !ok64)) ? 128 : (ok64 ? 64 : 0);
        } else if (!seq_halo_ok(p, bm)) return false;
        if (bm) {
            L.cfg = (signed char)(bm == 128 ? SEQ_CFG_HALO128 : SEQ_CFG_HALO64);
            L.wgt_frag = p.wgt_frag_halo;
        }
    }
    L.sync = 1;
    // K-loop stagger (smk_tune "seq_kstag": 0 off, 1 = layers whose weights fit the XCD's L2 beside the activations, 2 = all)
    L.kstag = (signed char)((g_tune.seq_kstag == 2 || (g_tune.seq_kstag == 1 && (size_t)p.Nst * p.Kpad * 2 <= (3u << 19))) ? 1 : 0);
    if (g_tune.seq_deep && L.cfg == 1) L.cfg = 5;         // measurement variant (smk_tune "seq_deep")
    {   // (smk_tune "seq_kstag_mask": which tile routines stagger -- 1 fused pairs, 2 patch-sharing tiles, 4 the im2col tiles)
        const bool is_halo = L.cfg == SEQ_CFG_HALO128 || L.cfg == SEQ_CFG_HALO64;
        if (is_halo && !(g_tune.seq_kstag_mask & 2)) L.kstag = 0;
        if (!is_halo && !(g_tune.seq_kstag_mask & 4)) L.kstag = 0;
    }
    return true;
}

// Pairs (conv3 of a Bottleneck, the 1x1 convolution that reads its output) -> one fused tile routine (c3c1_tile.inc): the 1x1
// needs every channel of a pixel and no neighbour, so the workgroup that owns 32 whole rows of conv3's output runs it from LDS.
// Marks the two records of every pair the routine has a shape for; the list itself (tensors, order, barriers behind the pair)
// stays as recorded.  smk_tune "seq_fuse" 0 leaves the list alone.
static bool seq_pair_fusable_why(const SeqLayer *L, int i, int *code, int *why);
static bool seq_pair_fusable(const SeqLayer *L, int i, int *code) {
    int why = 0;
    const bool ok = seq_pair_fusable_why(L, i, code, &why);
    // SMK_SEQ_DEBUG=1: why is a (1x1 + residual + ReLU, 1x1) pair of records NOT fused?  (stderr, once per list walk)
    if (!ok && why > 1 && getenv("SMK_SEQ_DEBUG"))
        fprintf(stderr, "[seq fuse] records %d, %d: not fusable, reason %d (Kpad %d Nst %d -> Nst %d, res %p relu %d, b.relu %d b.Ho %d Hs %d)\n", i, i + 1, why,
                (int)L[i].Kpad, (int)L[i].Nst, (int)L[i + 1].Nst, L[i].res, (int)L[i].relu, (int)L[i + 1].relu, (int)L[i + 1].Ho, (int)L[i + 1].Hs);
    return ok;
}
static bool seq_pair_fusable_why(const SeqLayer *L, int i, int *code, int *why) {
    const SeqLayer &a = L[i], &b = L[i + 1];
    auto plain1x1 = [](const SeqLayer &l) {
        return l.kh == 1 && l.kw == 1 && l.stride == 1 && l.stride_x == 1 && l.pad == 0 && l.org_y == 0 && l.org_x == 0 &&
               l.Hl == l.Hs && l.Wl == l.Ws && l.Ho == l.Hs && l.Wo == l.Ws && l.Ci == l.Kpad;
    };
    if (!plain1x1(a) || !a.sync) { *why = 1; return false; }
    if (!plain1x1(b)) { *why = 2; return false; }
    if (!a.res || a.res_mode != RES_PRE_RELU || !a.relu) { *why = 1; return false; }
    if (b.res || b.res_mode != RES_NONE) { *why = 3; return false; }
    if (b.in != a.out) { *why = 1; return false; }
    if (b.cin_off != a.cout_off || b.Cs != a.Cos || b.Ci != a.Nst || b.Hs != a.Ho || b.Ws != a.Wo) { *why = 4; return false; }
    if (b.out == a.out || b.out == a.res || b.out == a.in) { *why = 5; return false; }
    // The routine fetches the residual BEFORE it waits at its hoist point.  The barrier still pending there is the one behind
    // the LAST layer before i that carries one (`pend`; layer i - 1 when it has sync = 1, an earlier one when smk_op_conv_seq's
    // caller chained independent members with sync = 0).  Whoever wrote the residual inside this list must be separated from
    // layer i by a barrier the workgroup has already PASSED, i.e. one behind a layer j' with writer <= j' < pend.
    // (The first record of an already marked pair carries no barrier of its own.)
    auto has_bar = [&](int k) { return L[k].sync && L[k].cfg != SEQ_CFG_C3C1_L3 && L[k].cfg != SEQ_CFG_C3C1_L2 && L[k].cfg != SEQ_CFG_C3C1P_L3 && L[k].cfg != SEQ_CFG_C3C1P_L2; };
    int pend = -1;
    for (int k = i - 1; k >= 0; --k)
        if (has_bar(k)) { pend = k; break; }
    for (int j = i - 1; j >= 0; --j)
        if (L[j].out == a.res) {
            bool passed = false;
            for (int k = j; k < pend; ++k) passed = passed || has_bar(k);
            if (!passed) { *why = 6; return false; }
            break;
        }
    // conv3's own input must be behind the pending barrier too (the hoist point is the only wait in front of its loads)
    for (int j = i - 1; j >= 0; --j)
        if (L[j].out == a.in) {
            if (j > pend) { *why = 7; return false; }
            break;
        }
    if (a.Kpad == 256 && a.Nst == 1024 && b.Nst == 256) *code = SEQ_CFG_C3C1_L3;
    else if (a.Kpad == 128 && a.Nst == 512 && b.Nst == 128) *code = SEQ_CFG_C3C1_L2;
    else { *why = 8; return false; }
    return true;
}
static int g_seq_fused_last = 0;          // pairs fused in the list that was launched last (smk_tune_get "seq_fused_last", a diagnostic)
static void seq_fuse_pairs(SeqLayer *L, int n, int B, const std::vector<char> *locked = nullptr, bool have_xch = false) {
    g_seq_fused_last = 0;
    if (!g_tune.seq_fuse) return;
    for (int i = 0; i + 1 < n; ++i) {
        int code = 0;
        if (locked && ((*locked)[i] || (*locked)[i + 1])) continue;        // (per-op tests: the caller forced a tile)
        if (L[i].cfg >= SEQ_CFG_C3C1_L3 || L[i + 1].cfg >= SEQ_CFG_C3C1_L3 || !seq_pair_fusable(L, i, &code)) continue;
        if (g_tune.seq_fuse == 2 && code != SEQ_CFG_C3C1_L3) continue;      // (2: layer3's pairs only, A/B knob)
        // Measured (profiles/r03h_*): -4.3 .. -5.5 % on the B = 8 step, -2.0 % at B = 16, -2.7 % at B = 24.  (What looked like a race of
        // this routine at B = 12 was a buffer shared by two layouts inside the launch, see build_arena; profiles/r03h_b12_race.txt.)
        // the routine switches rows beyond the image off with a buffer offset of 0x7ffff000: every tensor must end below it
        const size_t px = (size_t)B * L[i].Ho * L[i].Wo;
        const size_t widest = std::max(std::max((size_t)L[i].Cs, (size_t)L[i].Cos), std::max((size_t)L[i].res_Cs, (size_t)L[i + 1].Cos));
        if (px * widest * 2 >= 0x7fff0000u || L[i].in_bytes >= 0x7fff0000u) continue;
        // smk_tune "seq_pair2d": the pair split over two CUs (needs the exchange scratch: f16 contexts / smk_op_conv_seq have it)
        if (have_xch && g_tune.seq_pair2d && (g_tune.seq_pair2d == 1 || code == SEQ_CFG_C3C1_L3))
            code = code == SEQ_CFG_C3C1_L3 ? SEQ_CFG_C3C1P_L3 : SEQ_CFG_C3C1P_L2;
        L[i].cfg = (signed char)code;
        L[i + 1].cfg = (signed char)SEQ_CFG_C3C1_2ND;
        if (!(g_tune.seq_kstag_mask & 1)) L[i].kstag = 0;
        ++g_seq_fused_last;
        ++i;
    }
}

// Resident trunk (round 6; smk_kernels.h SEQ_YRES_*): consecutive fused pairs of one ResNet layer -- [conv3 k + conv1 k+1], conv2 k+1 on a
// patch-sharing tile, [conv3 k+1 + conv1 k+2] -- run on the same 32-row tiles; with ONE image per team and a tile per workgroup the
// same workgroup owns the same rows in both, and the second pair's residual is the Y image the first one left in its LDS
// (experiments/siammask_sharp/resnet.py:80-103: `out += residual`, residual = the previous block's output).  Marks: the second
// pair does not fetch its residual rows (64 KB per CU "usually from beyond the L2", profiles/r05_seq_phase_clocks.txt: 3.0-3.5 of a
// layer3 pair's 17-18 us), the first one does not store Y when nobody else reads the tensor, the 3x3 convolution between them works
// in the LDS behind the image.  Values and summation orders are unchanged: bit-identical (tests/test_gpu_seq.py).
// `keep` (per-op tests): records whose output the caller reads back.
static int g_seq_yres_last = 0;           // pairs that found their residual resident in the list launched last (smk_tune_get "seq_yres_last")
static void seq_mark_resident(SeqLayer *L, int n, int B, int nslots, const void *extern_read = nullptr, const std::vector<char> *keep = nullptr) {
    g_seq_yres_last = 0;
    if (!g_tune.seq_yres || B > 8) return;               // (image b runs on team b % 8: from nine images on a workgroup owns two tiles per pair)
    int prev = -1;
    for (int i = 0; i + 1 < n; ++i) {
        const int cfg = L[i].cfg;
        if (cfg != SEQ_CFG_C3C1_L3 && cfg != SEQ_CFG_C3C1_L2) continue;
        const int p = prev;
        prev = i;
        if (p < 0 || L[p].cfg != cfg || i != p + 3) continue;
        const int mid = L[p + 2].cfg;
        if (mid != SEQ_CFG_HALO64 && mid != SEQ_CFG_HALO128) continue;
        if ((L[i].Ho * L[i].Wo + 31) / 32 > nslots || L[i].Ho != L[p].Ho || L[i].Wo != L[p].Wo) continue;
        if (L[i].res != L[p].out || L[i].res_Cs != L[p].Cos || L[i].res_coff != L[p].cout_off) continue;
        L[i].a_stage |= SEQ_YRES_IN;
        L[p + 2].a_stage |= SEQ_LDS_HI;
        ++g_seq_yres_last;
        // the store of Y: only the pair's own second record (from LDS) and this residual read the tensor?
        bool others = L[p].out == extern_read || (keep && (*keep)[p]);
        for (int j = 0; j < n && !others; ++j) {
            if (j != p + 1 && L[j].in == L[p].out) others = true;
            if (j != i && L[j].res == L[p].out) others = true;
        }
        if (!others) L[p].a_stage |= SEQ_YRES_NOSTORE;
    }
}

// Triples (round 4): [conv2 (3x3, stride 1, pad = dilation), conv3, the next 1x1] of a Bottleneck as ONE tile routine on image-row
// tiles (c3c1_tile.inc, FRONT = 1).  Runs behind seq_fuse_pairs: a marked pair (i + 1, i + 2) whose first record reads what record i --
// the block's 3x3 convolution -- writes, and nobody else reads it.  The barrier between conv2 and the pair disappears with the
// tensor.  wstd[i] = the (kh, kw, cin)-ordered fragment pack of record i (the record itself carries the chunk-major pack of the
// patch-sharing tile).  smk_tune "seq_fuse3": 0 off, 1 on, 2 layer3's blocks only.
static int g_seq_fused3_last = 0;
static void seq_fuse_triples(SeqLayer *L, int n, int B, const void *const *wstd, const std::vector<char> *locked = nullptr) {
    g_seq_fused3_last = 0;
    if (!g_tune.seq_fuse3 || !wstd) return;
    auto group_has_bar = [&](int k) {            // a barrier stands behind record k (pairs / triples: behind their LAST record only)
        const int cf = L[k].cfg;
        if (cf == SEQ_CFG_C3C1_L3 || cf == SEQ_CFG_C3C1_L2 || cf == SEQ_CFG_C3C1P_L3 || cf == SEQ_CFG_C3C1P_L2 || cf == SEQ_CFG_C2C3C1_L3 ||
            cf == SEQ_CFG_C2C3C1_L2 || cf == SEQ_CFG_C2C3C1_MID)
            return false;
        return L[k].sync != 0;
    };
    for (int i = 0; i + 2 < n; ++i) {
        SeqLayer &c2 = L[i], &c3 = L[i + 1], &c1 = L[i + 2];
        if (locked && ((*locked)[i] || (*locked)[i + 1] || (*locked)[i + 2])) continue;
        if ((c3.cfg != SEQ_CFG_C3C1_L3 && c3.cfg != SEQ_CFG_C3C1_L2) || c1.cfg != SEQ_CFG_C3C1_2ND) continue;
        if (c2.cfg != SEQ_CFG_HALO128 && c2.cfg != SEQ_CFG_HALO64 && c2.cfg > 9) continue;     // (a plain tile or the patch-sharing one)
        const int code = c3.cfg == SEQ_CFG_C3C1_L3 ? SEQ_CFG_C2C3C1_L3 : SEQ_CFG_C2C3C1_L2;
        if (g_tune.seq_fuse3 == 2 && code != SEQ_CFG_C2C3C1_L3) continue;
        const int kc = c3.Kpad;                  // 256 / 128: conv2 is kc -> kc
        if (c2.kh != 3 || c2.kw != 3 || c2.stride != 1 || c2.stride_x != 1 || c2.pad != c2.dil || c2.dil < 1 || c2.dil > 2) continue;
        if (c2.Ci != kc || c2.Nst != kc || c2.Kpad != 9 * kc || !c2.relu || c2.res || c2.res_mode != RES_NONE || !c2.sync) continue;
        if (c2.org_y || c2.org_x || c2.Hl != c2.Hs || c2.Wl != c2.Ws || c2.Ho != c2.Hs || c2.Wo != c2.Ws) continue;
        if (c2.Wo > 32 || c2.Wo < 24 || c2.Wo + 2 * c2.dil > 35) continue;      // one image row per 32-row tile; short rows (the template's 15 x 15) stay pairs
        if (c3.in != c2.out || c3.cin_off != c2.cout_off || c3.Cs != c2.Cos || c3.Hs != c2.Ho || c3.Ws != c2.Wo) continue;
        if (!wstd[i]) continue;
        bool other_reader = false;               // conv2's output never reaches memory: nobody else may read it ...
        for (int j = i + 2; j < n; ++j) {        // ... until a later record writes that buffer again (the blocks of a layer share their intermediates)
            if (L[j].in == c2.out || L[j].res == c2.out) { other_reader = true; break; }
            if (L[j].out == c2.out) break;
        }
        if (other_reader || c2.out == c3.res || c2.out == c1.out || c2.out == c3.out) continue;
        // conv2's input must have been written in front of the barrier this routine waits for (the last one before record i)
        int pend = -1;
        for (int k = i - 1; k >= 0; --k)
            if (group_has_bar(k)) { pend = k; break; }
        bool in_ok = true;
        for (int j = i - 1; j >= 0; --j)
            if (L[j].out == c2.in) { in_ok = j <= pend; break; }
        if (!in_ok) continue;
        // the residual rows: in front of the wait when their writer is separated from record i by a barrier ALREADY passed, or when it
        // is the previous triple's conv3 on the same row tiles (then this very workgroup wrote them); behind the wait otherwise
        int res_late = 0;
        for (int j = i - 1; j >= 0; --j)
            if (L[j].out == c3.res) {
                bool passed = false;
                for (int k = j; k < pend; ++k) passed = passed || group_has_bar(k);
                const bool own_rows = L[j].cfg == SEQ_CFG_C2C3C1_MID && L[j].Ho == c3.Ho && L[j].Wo == c3.Wo;
                if (!passed && !own_rows) res_late = 1;
                break;
            }
        c2.cfg = (signed char)code;
        c2.wgt_frag = wstd[i];
        c2.sync = 0;
        c3.cfg = (signed char)SEQ_CFG_C2C3C1_MID;
        c3.a_stage = (signed char)res_late;
        ++g_seq_fused3_last;
        i += 2;
    }
}

// SMK_SEQ_CLK (measurement aid, eager runs only): print what (team 0, slot 0) stamped
static void seq_print_clk(const SeqArgs &a, const std::vector<std::string> &ids, const char *idn, const unsigned long long *h,
                          const unsigned long long *h2) {
    fprintf(stderr, "[seq clk] %s total %.2f us\n", idn, (h[2 * a.n] - h[0]) / 100.0);
    for (int i = 0; i < a.n; ++i)
        fprintf(stderr, "[seq clk]   %-10s cfg %d sync %d kstag %d  tiles %.2f us  arrive %.2f us\n", ids[i].c_str(), a.L[i].cfg,
                a.L[i].sync, a.L[i].kstag, (h[1 + 2 * i] - h[2 * i]) / 100.0, (h[2 + 2 * i] - h[1 + 2 * i]) / 100.0);
    if (!h2) return;
    // arrival of team 0's 32 workgroups at the barrier behind each layer (SMK_SEQ_CLK=2): how much of a team wait is SKEW (the last
    // arrival against the others) and how much the barrier's own latency (last arrival -> slot 0 past its wait in the next layer)?
    for (int i = 0; i < a.n; ++i) {
        const unsigned long long *ar = h2 + 12 * SEQ_MAX + 32 * i;
        unsigned long long mn = ~0ull, mx = 0;
        int nz = 0;
        double sum = 0.0;
        for (int q = 0; q < 32; ++q)
            if (ar[q]) { mn = std::min(mn, ar[q]); mx = std::max(mx, ar[q]); ++nz; }
        if (nz < 2) continue;
        for (int q = 0; q < 32; ++q)
            if (ar[q]) sum += (double)(mx - ar[q]);
        unsigned long long rel = 0;                      // slot 0 past the wait: stamp 7 of its first tile in the next record that has one
        for (int j = i + 1; j < a.n && !rel; ++j) rel = h2[12 * j + 7];
        fprintf(stderr, "[seq arrive] %-10s %2d workgroups: first -> last arrival %.2f us, mean wait for the last one %.2f us, last arrival -> slot 0 "
                "released %.2f us\n", ids[i].c_str(), nz, (mx - mn) / 100.0, sum / nz / 100.0, rel > mx ? (rel - mx) / 100.0 : -1.0);
    }
    for (int i = 0; i < a.n; ++i) {
        const unsigned long long *t = h2 + 12 * i;
        if (!t[0] || !t[6]) continue;
        const double us = (t[6] - t[0]) / 100.0;
        if (a.L[i].cfg == SEQ_CFG_C3C1P_L3 || a.L[i].cfg == SEQ_CFG_C3C1P_L2) {     // a pair split over two CUs: c3c1p_tile's phases
            fprintf(stderr, "[seq clk2]  %-10s first tile (pair split, fused with the next 1x1): team wait %.2f | activation rows -> LDS %.2f | "
                    "conv3 K loop %.2f | residual -> Y %.2f | Y = relu(..) %.2f | Y stores + second K loop + slab + arrive %.2f | partner wait + add %.2f | "
                    "stores %.2f us | %.0f MHz\n",
                    ids[i].c_str(), (t[7] - t[0]) / 100.0, (t[1] - t[7]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0,
                    (t[4] - t[3]) / 100.0, (t[5] - t[4]) / 100.0, (t[10] - t[5]) / 100.0, (t[6] - t[10]) / 100.0, us > 0 ? (double)(t[9] - t[8]) / us : 0.0);
            continue;
        }
        if (a.L[i].cfg == SEQ_CFG_C2C3C1_L3 || a.L[i].cfg == SEQ_CFG_C2C3C1_L2) {   // a triple: c3c1_tile's phases with conv2 in front
            fprintf(stderr, "[seq clk2]  %-10s first tile (3x3 + conv3 + the next 1x1): prologue + team wait %.2f | patch -> LDS + conv2 %.2f | "
                    "conv3 K loop %.2f | residual -> Y %.2f | Y = relu(..) %.2f | Y stores + second K loop %.2f | its epilogue + stores %.2f us | %.0f MHz\n",
                    ids[i].c_str(), (t[7] - t[0]) / 100.0, (t[1] - t[7]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0,
                    (t[4] - t[3]) / 100.0, (t[5] - t[4]) / 100.0, (t[6] - t[5]) / 100.0, us > 0 ? (double)(t[9] - t[8]) / us : 0.0);
            continue;
        }
        if (a.L[i].cfg == SEQ_CFG_C3C1_L3 || a.L[i].cfg == SEQ_CFG_C3C1_L2) {       // a fused pair: c3c1_tile's phases
            fprintf(stderr, "[seq clk2]  %-10s first tile (fused with the next 1x1): team wait %.2f | activation rows -> LDS %.2f | "
                    "conv3 K loop %.2f | residual -> Y %.2f | Y = relu(..) %.2f | Y stores + second K loop %.2f | its epilogue + stores %.2f us | %.0f MHz\n",
                    ids[i].c_str(), (t[7] - t[0]) / 100.0, (t[1] - t[7]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0,
                    (t[4] - t[3]) / 100.0, (t[5] - t[4]) / 100.0, (t[6] - t[5]) / 100.0, us > 0 ? (double)(t[9] - t[8]) / us : 0.0);
            continue;
        }
        fprintf(stderr, "[seq clk2]  %-10s first tile: prologue %.2f | team wait %.2f | first operands %.2f | K loop %.2f | other waves %.2f | "
                "acc -> LDS %.2f | bias/res/stores %.2f | end sync %.2f us | %.0f MHz\n", ids[i].c_str(),
                t[7] ? (t[7] - t[0]) / 100.0 : 0.0, 0.0, t[7] ? (t[1] - t[7]) / 100.0 : (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0,
                (t[3] - t[2]) / 100.0, (t[4] - t[3]) / 100.0, (t[5] - t[4]) / 100.0, (t[6] - t[5]) / 100.0,
                us > 0 ? (double)(t[9] - t[8]) / us : 0.0);
    }
}

static int seq_flush(smk_ctx *c, int B, hipStream_t s) {
    const size_t n = c->seq_rec.size();
    for (size_t i0 = 0; i0 < n; i0 += SEQ_MAX) {
        SeqArgs a;
        memset(&a, 0, sizeof(a));
        a.n = (int)(n - i0 < (size_t)SEQ_MAX ? n - i0 : SEQ_MAX);
        a.B = B;
        a.flags = g_tune.seq_spoll ? 1 : 0;
        a.bar = c->seq_bar;
        a.xch = c->seq_xch;
        a.err = c->seq_err;
        a.err_host = c->seq_err_hdev;
        // pipelined step, depth 2: the last launch of the list tells the previous frame's chain / mask-head launch that the chip is free
        a.exit_sem = (c->pipe_seq_exit && i0 + SEQ_MAX >= n) ? c->pipe_cnt + 7 : nullptr;
        if (a.exit_sem) c->pipe_seq_exit_done = true;
        for (int i = 0; i < a.n; ++i) a.L[i] = c->seq_rec[i0 + i];
        seq_fuse_pairs(a.L, a.n, B, nullptr, c->seq_xch != nullptr && (c->seq_grid >> 3) % 2 == 0 && (c->seq_grid >> 4) <= SEQ_XCH_PAIRS);   // (a pair never straddles two launches)
        seq_fuse_triples(a.L, a.n, B, c->seq_wstd.data() + i0);
        seq_mark_resident(a.L, a.n, B, c->seq_grid >> 3, c->buf.count("p2") ? c->buf.at("p2") : nullptr);
        const char *ck = getenv("SMK_SEQ_CLK");
        const bool want_clk = ck != nullptr && !c->graph_mode;
        // SMK_SEQ_CLK=2: additionally the phases INSIDE the first tile of every layer (a separate kernel build with the stamps)
        const bool want_clk2 = want_clk && !strcmp(ck, "2");
        if (want_clk && !c->seq_clk) HIPCHK(hipMalloc((void **)&c->seq_clk, sizeof(unsigned long long) * (2 * SEQ_MAX + 1)));
        if (want_clk2 && !c->seq_clk2) HIPCHK(hipMalloc((void **)&c->seq_clk2, sizeof(unsigned long long) * (12 + 32) * SEQ_MAX));
        if (want_clk2) HIPCHK(hipMemsetAsync(c->seq_clk2, 0, sizeof(unsigned long long) * (12 + 32) * SEQ_MAX, s));
        a.clk = want_clk ? c->seq_clk : nullptr;
        a.clk2 = want_clk2 ? c->seq_clk2 : nullptr;
        char idn[96];
        snprintf(idn, sizeof(idn), "seq[%s..%s]", c->seq_ids[i0].c_str(), c->seq_ids[i0 + a.n - 1].c_str());
        const double fr = (double)a.n / (double)n;
        ProfScope ps(c, s, idn, "conv_seq", c->seq_flop * fr, c->seq_bytes * fr);
        {   // what must cross the fabric if every tensor produced AND consumed inside the launch stays in the XCD's L2:
            // tensors read but not produced here, every weight pack once, tensors produced here and not read here (+ p2,
            // which Refine reads later)
            double ext = 0.0;
            for (int i = 0; i < a.n; ++i) {
                const SeqLayer &L = a.L[i];
                bool in_inside = false, res_inside = L.res == nullptr, out_read = false;
                for (int j = 0; j < a.n; ++j) {
                    if (j < i && a.L[j].out == L.in) in_inside = true;
                    if (j < i && L.res && a.L[j].out == L.res) res_inside = true;
                    if (j > i && (a.L[j].in == L.out || a.L[j].res == L.out)) out_read = true;
                }
                bool in_counted = false, res_counted = false;          // a tensor several layers read is fetched once
                for (int j = 0; j < i; ++j) {
                    if (a.L[j].in == L.in || a.L[j].res == L.in) in_counted = true;
                    if (L.res && (a.L[j].in == L.res || a.L[j].res == L.res)) res_counted = true;
                }
                const double px_in = (double)B * L.Hs * L.Ws, px_out = (double)B * L.Ho * L.Wo;
                ext += (double)L.Nst * L.Kpad * 2.0;
                if (!in_inside && !in_counted) ext += px_in * L.Cs * 2.0;
                if (!res_inside && !res_counted) ext += px_out * L.res_Cs * 2.0;
                if (!out_read || L.out == c->buf.at("p2")) ext += px_out * L.Nst * 2.0;
            }
            ps.ext_bytes(ext);
        }
        if (launch_conv_seq(a, c->seq_grid, s))
            return fail(SMK_E_HIP, "launch of %s failed: %s", idn, hipGetErrorString(hipGetLastError()));
        c->seq_pending = c->cap_has_seq = true;          // (smk_seq_sync_check: the flag is worth a look once this has drained)
        if (want_clk) {                                  // per-layer spans of (team 0, slot 0), eager mode only
            unsigned long long h[2 * SEQ_MAX + 1], h2[(12 + 32) * SEQ_MAX];
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(h, c->seq_clk, sizeof(h), hipMemcpyDeviceToHost));
            if (want_clk2) HIPCHK(hipMemcpy(h2, c->seq_clk2, sizeof(h2), hipMemcpyDeviceToHost));
            std::vector<std::string> ids(c->seq_ids.begin() + i0, c->seq_ids.begin() + i0 + a.n);
            seq_print_clk(a, ids, idn, h, want_clk2 ? h2 : nullptr);
        }
    }
    c->seq_rec.clear();
    c->seq_wstd.clear();
    c->seq_ids.clear();
    c->seq_flop = c->seq_bytes = 0.0;
    return 0;
}

// The persistent kernel reports placement violations and barrier time-outs through a flag in host-mapped memory.  Every
// entry point reads it (a plain host load, no synchronisation): on failure the context stops using sequences, the team
// counters, the flags and every captured graph (they contain sequence launches) are reset, and the call fails with
// SMK_E_SEQ -- the results of the calls enqueued since the failure are not valid and the caller re-submits them.
static int seq_health(smk_ctx *c) {
    if (!c->seq_err_host) return 0;
    const int e = *(volatile int *)c->seq_err_host;
    if (!e) return 0;
    (void)hipDeviceSynchronize();                        // launches that found the flag set returned at once
    c->seq_fail = e;
    if (e != 3) c->seq_grid = 0;                         // (3: the pipelined step's gate timed out -- nothing wrong with the sequences)
    else c->pipe_depth = 0;                              // ... but something serialises this context's two queues (a counter-collecting profiler
                                                         // does): frame steps are serial from here on (smk_set_pipeline turns it back on)
    if (c->pipe_cnt) {                                   // the pipelined step's semaphores at rest (see pipe_reset_counters)
        const unsigned init[16] = {1u, 0u};
        (void)hipMemcpy(c->pipe_cnt, init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (c->pipe_sig) { (void)hipMemset(c->pipe_sig, 0, 8); c->pipe_sig_n = 0; }
    *(volatile int *)c->seq_err_host = 0;
    (void)hipMemset(c->seq_err, 0, sizeof(int));
    (void)hipMemset(c->seq_bar, 0, 8 * 32 * sizeof(unsigned));
    for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    c->graph_used.clear();
    c->graph_has_seq.clear();
    c->seq_pending = false;
    c->tail_pending = false;                             // (the device has drained)
    c->tail2_pending = false;                            // (a pending second tail part belongs to an invalid frame: dropped with its graph)
    c->template_B = 0;                                   // nothing says the cached template features were computed before the failure
    c->track_B = 0;
    if (e == 3)
        return fail(SMK_E_SEQ, "conv_seq_kernel reported: (pipelined step) a gate waited 0.2 s (a tail's gate: 5 s) for its partner on the other stream -- something serialises "
                    "the two queues of this context (a profiler collecting counters does); the results of the calls enqueued on this context "
                    "since then are invalid (the cached template included) and frame steps are serial from here on: call template() again and "
                    "re-submit the frame");
    return fail(SMK_E_SEQ, "conv_seq_kernel reported %s: the results of the calls enqueued on this context since then are invalid "
                "(the cached template included); persistent sequences are now off for this context (per-layer kernels from here "
                "on): call template() again and re-submit the frame",
                e == 1 ? "an uneven distribution of workgroups over the XCDs" : "a team-barrier time-out (another kernel held CUs "
                "for more than 0.2 s, or a second persistent kernel ran beside it)");
}

static bool seq_wanted(const smk_ctx *c, int B) {
    // Image b runs on XCD b % 8, so the sequence pays when the XCDs are (nearly) all busy and evenly loaded.  Measured with
    // the sequence forced on for every batch size against the per-launch path, same process (tools/measure/gpu_seq_batch_sweep.py,
    // profiles/r03_seq_batch_sweep.txt): B = 6 / 7 / 8 x1.02 / 1.05 / 1.10, B = 16 x1.07, B = 24 x1.02; B <= 4 x0.85-0.91 (idle
    // XCDs: the per-launch kernels spread an image over the chip), B = 5 and 12 x1.00, B = 10 x0.97 (two XCDs run two images),
    // B = 32 x0.95 (four images in sequence on 64-row tiles lose to the chip-wide 128 / 256-row tiles).
    if (!g_tune.seq || c->seq_grid <= 0 || c->dtype != DT_F16) return false;
    // End of round 3 (fused pairs, patch-sharing tiles: the sequence itself 13 % faster; profiles/r03h_seq_batch_sweep.txt): B = 5 x1.076 and
    // B = 12 x1.029 (seq_extra_batch) join; B = 3 / 4 x0.954 / 0.983, B = 10 x0.990, B = 32 x0.955 stay on the per-launch path.
    if (B >= g_tune.seq_min_batch && B <= g_tune.seq_max_batch) return true;
    if (B == g_tune.seq_extra_batch) return true;
    return B % 8 == 0 && B <= g_tune.seq_mult_max;
}

// conv_wreg_kernel (weights global -> VGPR) or the LDS-staged kernels?  Returns the tile code 1..6
// (64x256, 64x128, 64x64, 128x256, 128x128, 128x64) or 0.
// CUs of the current device; host-only callers (smk_host_plan_conv on a box without a GPU) plan for the MI355X's 256
static long device_cus() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 256; }
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        cached[dev] = n;
    }
    return cached[dev];
}
static const int WREG_TILE[9][2] = {{0, 0}, {64, 256}, {64, 128}, {64, 64}, {128, 256}, {128, 128}, {128, 64}, {96, 256}, {32, 64}};
static int wreg_choice(const ConvParams &p, const ConvOpt &o, int dtype) {
    if (o.algo_naive || !conv_wreg_eligible(p, dtype)) return 0;
    if (o.wreg) return o.wreg;
    if (g_tune.wreg >= 2) return g_tune.wreg - 1;
    if (!g_tune.wreg) return 0;
    // per-shape choice, fitted to profiles/r02_wregbench_b8_b64.json (A/B against the best LDS-staged instantiation in
    // one process): the register path wins where the weight stream is long and the tile is N-wide -- the two strided /
    // wide 3x3 projections (l2.0.ds x1.10-1.15, l3.0.ds x1.05-1.08) and, while M is small (B <= ~16), the 1x1
    // reductions with K >= 512 (l3.c1 x1.13, l3.0.c1 x1.10, l2.c1 x1.07) and the strided 3x3 of l2.0 (x1.06).
    // It loses on layer1 (short K, large M: x0.5-0.9) and against the halo kernel on 3x3 stride-1 layers.
    const long K = (long)p.kh * p.kw * p.Ci;
    if (g_tune.wreg_policy == 1) {
        // With four producer waves (smk_tune npw, round 2) the register-fed kernel beats the best LDS-staged instantiation on
        // almost every fp16 NHWC layer of the path at B = 1, 8 and 64 (profiles/r02_producer_waves_2_vs_4.txt,
        // r02_producer_waves_layers_b1_b64.json).  Exceptions, kept on the LDS-staged kernels: the 7x7 stem, the short-K 3x3
        // stride-1 layers (the patch-sharing kernel wins or ties: l1.c2, l2.c2, Refine's small convolutions), and at
        // large M the narrow / short-K layers (128-row LDS-staged tiles at two workgroups per CU win: layer1, l2.c1, head0,
        // v1.0 at B = 64).
        if (p.kh > 3) return 0;
        if (p.kh == 3 && p.stride == 1 && K <= 1152) return 0;
        if (p.M > 16384 && !(p.Nst >= 512 || (K >= 2304 && p.Nst >= 128) || (K >= 1024 && p.Nst >= 256) ||
                             (p.kh == 3 && p.stride == 2)))
            return 0;
        // the largest workgroup shape that still hands the chip >= 150 workgroups (N-wide first: 128x256, 64x256, 64x128, 64x64)
        static const int cand[4] = {4, 1, 2, 3};
        const int nr = (p.Nst + 63) / 64 * 64, ng = p.groups > 0 ? p.groups : 1;
        for (int ci = 0; ci < 4; ++ci) {
            const int bm = WREG_TILE[cand[ci]][0], bn = WREG_TILE[cand[ci]][1];
            if (bn > nr && bn > 64) continue;
            if ((long)((p.M + bm - 1) / bm) * ((p.Nst + bn - 1) / bn) * ng >= 150) {
                // 128 x 256 with 129 .. 255 workgroups leaves CUs idle for a whole tile time (conv_search at B = 8: 53 x 3 = 159
                // tiles on 256 CUs); 96 rows (code 7) = 213 tiles, still one round, each 3/4 as long (smk_tune "wreg96", round 5)
                if (cand[ci] == 4 && g_tune.wreg96) {
                    const long ncu = device_cus(), tn = (p.Nst + 255) / 256 * ng;
                    const long t128 = (long)((p.M + 127) / 128) * tn, t96 = (long)((p.M + 95) / 96) * tn;
                    if (((t128 + ncu - 1) / ncu) * 128 > ((t96 + ncu - 1) / ncu) * 96 && t96 <= 4 * ncu) return 7;
                }
                return cand[ci];
            }
        }
        // nothing fills the chip: 64 x 64 -- or 32 x 64 (code 8, smk_tune "wreg32") while even that leaves CUs idle: the narrow tiles' K loops wait for
        // their activation refills (profiles/r06t_wreg_ring_depth.txt), and a 32-row workgroup asks for half of them
        if (g_tune.wreg32 && (long)((p.M + 63) / 64) * ((p.Nst + 63) / 64) * ng < g_tune.wreg32) return 8;
        return 3;
    }
    if (p.M < 4096) return 0;                  // not measured below B ~ 5: keep the fitted LDS-staged choice
    if (p.kh == 3 && K >= 2304 && p.Nst >= 512)
        return ((long)((p.M + 127) / 128) * ((p.Nst + 255) / 256) >= 200) ? 4 : 1;       // 128x256 once it fills the chip
    if (p.M > 16384) return 0;
    if (p.kh == 1 && K >= 512 && p.Nst <= 256) return p.Nst >= 192 ? 2 : 3;               // 64x128 / 64x64
    if (p.kh == 3 && p.stride == 2 && K >= 1152 && p.Nst <= 128) return 3;
    return 0;
}
// depth of conv_wreg_kernel's activation ring: three K tiles; four in split-operand contexts (their K loops are three times as long and mostly on the
// narrow tiles: +2.3 .. 2.6 % on the B = 8 step) and for one or two streams (every layer on 64 x 64 tiles: +1.8 % on the B = 1 step).  Deeper rings (5 .. 7)
// lose everywhere, and so does running the WEIGHT stream further ahead: profiles/r06q_x3_ring_depth_b1_levers.txt, r06t_wreg_ring_depth.txt, r06r_wreg_deep_prefetch.txt
static int wreg_stages(const smk_ctx *c = nullptr, int B = 0) {
    if (g_tune.wreg_stages) return g_tune.wreg_stages;
    return (c && (c->dtype == DT_F16X3 || (B >= 1 && B <= 2))) ? 4 : 3;
}
// per-op entry points: bits 6-7 of the tile code -- 0 the library's choice, 1 eight k-steps ahead on every shape (conv_wreg.hip WregDepth; MEASURE builds,
// otherwise the three-deep ring), 2 / 3 a 3- / 4-deep ring
static int wreg_stages_from_code(int code) {
    const int st = (code >> 6) & 3;
    return st == 0 ? wreg_stages() : (st == 1 ? 8 : (st == 2 ? 3 : 4));
}

// conv_pp_kernel (conv_pp.hip: 256 x 256 tiles, two wave groups alternating between fetching and multiplying) takes the long-K
// convolutions once 256-row tiles fill the chip in (nearly) whole rounds -- the 3x3 shortcuts and layer3's conv2 from B ~ 53 (BASELINE
// configs[4]): +1.3..3.8 % per launch over the register-fed 128 x 256 tile; conv_search's 633 tiles are 2.47 rounds (0.82 of three) and
// stay on the register-fed kernel (-6 %); profiles/r06a_pp_first_contact.txt.  Below ~200 tiles the launch is a partial round of a few
// long tiles and the smaller tiles win (B = 32: -6..-60 %).  pp = 2 (A/B knob): any K, e.g. the Bottlenecks' 1x1 convolutions.
static bool pp_choice(const ConvParams &p, const ConvOpt &o, int dtype) {
    if (!g_tune.pp || o.algo_naive || o.halo || o.wreg || o.tile_code || !conv_pp_eligible(p, dtype)) return false;
    const long K = (long)p.kh * p.kw * p.Ci;
    const long tiles = (long)((p.M + 255) / 256) * ((p.Nst + 255) / 256), ncu = device_cus();
    const long rounds = (tiles + ncu - 1) / ncu;
    if (tiles < 200 || tiles * 10 < rounds * ncu * 9 || p.Nst < 256 || (p.Nst % 256) != 0) return false;
    return g_tune.pp == 2 || K >= 2304;
}

static int run_conv(smk_ctx *c, const char *id, const Act &in, const Act *out, int B, const ConvOpt &o,
                    hipStream_t s) {
    auto it = c->conv.find(id);
    if (it == c->conv.end()) return fail(SMK_E_STATE, "internal: conv %s not packed", id);
    ConvParams p;
    CHK(conv_params(c, it->second, in, out, B, o, p));
    const TileChoice t = tile_from_code(o.tile_code, p, kdtype(c->dtype));
    const int ng = p.groups > 0 ? p.groups : 1;
    // algorithmic work: 2*M*N*K_real flops; bytes = activations read once + weights + output written once
    const double kreal = it->second.alg_k ? (double)it->second.alg_k : (double)p.kh * p.kw * p.Ci;
    const double flop = 2.0 * p.M * (double)p.N * kreal * ng;
    const size_t es = esize(kdtype(c->dtype));
    const double in_bytes = (double)B * (p.ups ? p.Hs * p.Ws : (double)p.Hl * p.Wl) * p.Ci * es * ng;
    const double out_bytes = (double)p.M * p.N * ng * (p.out_mode == OUT_NCHW_F32 ? 4 : es);
    const double bytes = in_bytes + out_bytes + (double)p.N * kreal * es * ng + (p.res ? (double)p.M * p.N * es : 0.0);
    if (c->seq_on) {
        SeqLayer L;
        if (!o.algo_naive && !o.halo && !o.wreg && !o.tile_code && seq_layer_from(p, kdtype(c->dtype), L)) {
            c->seq_rec.push_back(L);
            c->seq_wstd.push_back(p.wgt_frag);
            c->seq_ids.push_back(id);
            c->seq_flop += flop;
            c->seq_bytes += bytes;
            return 0;
        }
        CHK(seq_flush(c, B, s));               // not eligible: keep program order
    }
    char kn[64];
    snprintf(kn, sizeof(kn), "conv_igemm<%s,%dx%dx%d,s%d,%s>", dtname(kdtype(c->dtype)), t.bm, t.bn, t.kt, t.stages,
             p.out_mode == OUT_NCHW_F32 ? "nchw" : "nhwc");
    int rc = 1;
    int bm = o.algo_naive ? 0 : halo_choice(it->second, p, o, kdtype(c->dtype));
    if (bm && !o.halo && conv_ksplit(p, kdtype(c->dtype), t) > 1) bm = 0;       // under-filled: split-K on the generic kernel wins
    // (policy 1 decides between the register-fed and the patch-sharing kernel itself; the round-2 table only covered the
    //  layers the patch-sharing kernel does not take)
    if (pp_choice(p, o, kdtype(c->dtype))) {
        ProfScope ps(c, s, id, "conv_pp<f16,256x256>", flop, bytes);
        rc = launch_conv_pp(p, s);
        if (rc == 1) ps.cancel();
        else bm = 0;
    }
    const int wr = (rc != 1 || o.halo || (bm && !o.wreg && g_tune.wreg < 2 && g_tune.wreg_policy == 0)) ? 0 : wreg_choice(p, o, kdtype(c->dtype));
    if (wr) {
        // weights straight into registers, activations through LDS
        ConvBatch cb;
        cb.n = 1;
        cb.p[0] = p;
        char kw_[64];
        snprintf(kw_, sizeof(kw_), "conv_wreg<%s,%dx%d,s%d>", dtname(kdtype(c->dtype)), WREG_TILE[wr][0], WREG_TILE[wr][1], wreg_stages(c, B));
        ProfScope ps(c, s, id, kw_, flop, bytes);
        rc = launch_conv_wreg_batch(cb, WREG_TILE[wr][0], WREG_TILE[wr][1], wreg_stages(c, B), s);
        if (rc == 1) ps.cancel();
        else bm = 0;
    }
    if (bm && rc == 1) {
        // 3x3 stride-1: the activation patch is staged once per channel chunk and shared by the nine taps
        ConvParams ph = p;
        ph.wgt = it->second.w_halo;
        char kh_[64];
        snprintf(kh_, sizeof(kh_), "conv3x3_halo<%s,%dx128>", dtname(kdtype(c->dtype)), bm);
        ProfScope ps(c, s, id, kh_, flop, bytes);
        rc = launch_conv_halo(ph, kdtype(c->dtype), bm, s);
        if (rc == 1) ps.cancel();
    }
    if (rc == 1) {
        ProfScope ps(c, s, id, o.algo_naive ? "conv_naive" : kn, flop, bytes);
        rc = o.algo_naive ? launch_conv_naive(p, kdtype(c->dtype), s) : launch_conv_mfma(p, kdtype(c->dtype), t, s);
    }
    if (rc) return fail(SMK_E_HIP, "launch of conv %s failed: %s", id, hipGetErrorString(hipGetLastError()));
    return 0;
}

// several independent convolutions as ONE launch (same dtype / epilogue mode / tile for all):
// removes launch boundaries and fills the chip when the single problems are small
struct ConvJob { const char *id; const Act *in; const Act *out; ConvOpt o; };

static int run_conv_jobs(smk_ctx *c, const std::vector<ConvJob> &jobs, int B, int lead, hipStream_t s) {
    if (jobs.empty() || (int)jobs.size() > CONV_BATCH_MAX) return fail(SMK_E_ARG, "internal: bad conv job count");
    if (c->seq_on) {
        // independent members: no team barrier between them, one after the last
        const size_t n0 = c->seq_rec.size();
        for (auto &j : jobs) CHK(run_conv(c, j.id, *j.in, j.out, B, j.o, s));
        if (c->seq_rec.size() == n0 + jobs.size())
            for (size_t i = n0; i + 1 < c->seq_rec.size(); ++i) c->seq_rec[i].sync = 0;
        return 0;
    }
    ConvBatch cb;
    cb.n = (int)jobs.size();
    // merging pays while the single problems under-fill the chip; once every halo-eligible member is a full
    // launch of BM=128 tiles on its own, separate halo launches are faster than the merged generic one
    bool split_for_halo = g_tune.halo != 0;
    int n_halo = 0;
    for (int i = 0; i < cb.n; ++i) {
        auto it = c->conv.find(jobs[i].id);
        if (it == c->conv.end()) return fail(SMK_E_STATE, "internal: conv %s not packed", jobs[i].id);
        CHK(conv_params(c, it->second, *jobs[i].in, jobs[i].out, B, jobs[i].o, cb.p[i]));
        const int bm = halo_choice(it->second, cb.p[i], jobs[i].o, kdtype(c->dtype));
        if (bm) ++n_halo;
        if (bm == 64 || (bm == 0 && cb.p[i].kh == 3)) split_for_halo = false;
    }
    // Merging is for launches whose members under-fill the chip.  At large batches every member is a full launch on its own and
    // the merged launch only forces ONE kernel instantiation on all of them (the shortcut 3x3 beside conv1 on the LDS-staged
    // 256x128 tile instead of the register-fed 128x256: B = 64 +5.1 % without merging, profiles/r04j_b64_merge_ab.txt).
    const bool merge_ok = g_tune.merge && (g_tune.merge == 2 || B <= g_tune.merge_max_batch);
    if ((c->prof && !c->prof_merge) || cb.n == 1 || !merge_ok || (split_for_halo && n_halo)) {   // per-layer attribution while profiling (mode 1)
        for (auto &j : jobs) CHK(run_conv(c, j.id, *j.in, j.out, B, j.o, s));
        return 0;
    }
    // (profile mode 2: one record for the merged launch, algorithmic work summed over its members as run_conv counts it)
    double mflop = 0.0, mbytes = 0.0;
    std::string mid;
    if (c->prof) {
        const size_t es = esize(kdtype(c->dtype));
        for (int i = 0; i < cb.n; ++i) {
            const ConvParams &q = cb.p[i];
            const PackedConv &pc = c->conv.find(jobs[i].id)->second;
            const int ng = q.groups > 0 ? q.groups : 1;
            const double kreal = pc.alg_k ? (double)pc.alg_k : (double)q.kh * q.kw * q.Ci;
            mflop += 2.0 * q.M * (double)q.N * kreal * ng;
            mbytes += (double)B * (q.ups ? q.Hs * q.Ws : (double)q.Hl * q.Wl) * q.Ci * es * ng +
                      (double)q.M * q.N * ng * (q.out_mode == OUT_NCHW_F32 ? 4 : es) + (double)q.N * kreal * es * ng +
                      (q.res ? (double)q.M * q.N * es : 0.0);
            mid += (i ? "+" : "") + std::string(jobs[i].id);
        }
    }
    {
        int wr = wreg_choice(cb.p[lead], jobs[lead].o, kdtype(c->dtype));
        for (int i = 0; i < cb.n && wr; ++i)
            if (!wreg_choice(cb.p[i], jobs[i].o, kdtype(c->dtype))) wr = 0;
        if (wr) {
            char kw_[64];
            snprintf(kw_, sizeof(kw_), "conv_wreg<%s,%dx%d,s%d,merged%d>", dtname(kdtype(c->dtype)), WREG_TILE[wr][0], WREG_TILE[wr][1], wreg_stages(c, B), cb.n);
            ProfScope ps(c, s, mid.c_str(), kw_, mflop, mbytes);
            const int rc = launch_conv_wreg_batch(cb, WREG_TILE[wr][0], WREG_TILE[wr][1], wreg_stages(c, B), s);
            if (rc == 0) return 0;
            ps.cancel();
            if (rc != 1) return fail(SMK_E_HIP, "launch of merged conv %s.. failed: %s", jobs[0].id, hipGetErrorString(hipGetLastError()));
        }
    }
    const TileChoice t = tile_from_code(jobs[lead].o.tile_code, cb.p[lead], kdtype(c->dtype));
    char km_[72];
    snprintf(km_, sizeof(km_), "conv_igemm<%s,%dx%dx%d,s%d,%s,merged%d>", dtname(kdtype(c->dtype)), t.bm, t.bn, t.kt, t.stages,
             cb.p[lead].out_mode == OUT_NCHW_F32 ? "nchw" : "nhwc", cb.n);
    ProfScope psm(c, s, mid.c_str(), km_, mflop, mbytes);
    if (launch_conv_mfma_batch(cb, kdtype(c->dtype), t, s))
        return fail(SMK_E_HIP, "launch of merged conv %s.. failed: %s", jobs[0].id, hipGetErrorString(hipGetLastError()));
    return 0;
}

// A Bottleneck's conv3 (+ residual + ReLU) and the 1x1 convolution that reads its output as ONE launch (conv_pair_kernel: the
// sequence's c3c1_tile per 32 rows of the flattened batch) -- for the batches that do not run the persistent sequence.
// Returns 1 when the pair is not eligible (the caller then issues the two convolutions), 0 when it was launched.
static int run_conv_pair(smk_ctx *c, const char *id3, const Act &in3, const Act &out3, const ConvOpt &o3, const char *id1,
                         const Act &out1, const ConvOpt &o1, int B, hipStream_t s) {
    if (c->dtype != DT_F16 || !g_tune.pair_launch || c->seq_on || parallel_ok(c) || (c->prof && !c->prof_merge)) return 1;
    // Measured (profiles/r04h_pair_launch_ab.txt, r04i_pair64_ab.txt; whole step, one process per batch, off / on / off / on):
    //   32-row tiles (c3c1_tile):  B = 1 +2 %, B = 4 -1.0 %, B = 10 -5..8 %, B = 32 0, B = 64 +1.2 %
    //   64-row tiles (c3c1s_tile): B = 1 +7 %, B = 10 -10..11 %, B = 32 -5.0 %, B = 64 -5.2 %
    // At B = 1 a pair is 31 (16) workgroups that each stream the full 1 MB of weights where the two launches spread the image over
    // the chip; from ~150 tiles on the 64-row form wins everywhere (half the weight bytes per row, the trunk written once and
    // never re-read).  1 = the rule (off for B <= 2, 32-row tiles for 3 <= B <= 8, 64-row tiles from B = 9), 2 / 3 = always 32 / 64 rows.
    if (g_tune.pair_launch == 1 && B < 3) return 1;
    const int pair_rows = g_tune.pair_launch == 3 ? 64 : (g_tune.pair_launch == 2 ? 32 : (B >= 9 ? 64 : 32));
    auto i3 = c->conv.find(id3), i1 = c->conv.find(id1);
    if (i3 == c->conv.end() || i1 == c->conv.end() || o1.win || o1.ups || o1.pos) return 1;
    ConvParams p3, p1;
    if (conv_params(c, i3->second, in3, &out3, B, o3, p3) || conv_params(c, i1->second, out3, &out1, B, o1, p1)) return 1;
    SeqLayer L[2];
    if (!seq_layer_from(p3, c->dtype, L[0], -1) || !seq_layer_from(p1, c->dtype, L[1], -1)) return 1;
    int code = 0;
    if (!seq_pair_fusable(L, 0, &code)) return 1;
    const size_t px = (size_t)p3.M;
    const size_t widest = std::max(std::max((size_t)L[0].Cs, (size_t)L[0].Cos), std::max((size_t)L[0].res_Cs, (size_t)L[1].Cos));
    if (px * widest * 2 >= 0x7fff0000u || L[0].in_bytes >= 0x7fff0000u) return 1;     // (the routine's out-of-range offset: see seq_fuse_pairs)
    const size_t es = esize(c->dtype);
    const double flop = 2.0 * p3.M * ((double)p3.N * p3.Ci + (double)p1.N * p1.Ci);
    const double bytes = ((double)p3.M * (p3.Ci + 2.0 * p3.N + p1.N) + (double)p3.N * p3.Ci + (double)p1.N * p1.Ci) * es;
    char kn[48];
    snprintf(kn, sizeof(kn), "conv_pair<f16,%d-%d-%d>", p3.Ci, p3.N, p1.N);
    (void)pair_rows;
    const std::string pid = std::string(id3) + "+" + id1;
    ProfScope ps(c, s, pid.c_str(), kn, flop, bytes);
    if (launch_conv_pair(L[0], L[1], code, p3.M, s, pair_rows)) return fail(SMK_E_HIP, "launch of pair %s failed: %s", pid.c_str(), hipGetErrorString(hipGetLastError()));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// the network
// ---------------------------------------------------------------------------------------------
// modified ResNet-50 + adjust (resnet.py:217-227, custom.py:19-25,58-66) on an S x S input.
// S=255: leaves p0,p1,p2 (kept for Refine) and "search" [31,31,256];  S=127: leaves "zf" [7,7,256].
// phase: PH_FRONT = stem + maxpool + layer1 (p0, p1), PH_BACK = layer2 .. adjust from the kept p1; both = the whole backbone.
// The pipelined frame step (smk_set_pipeline) runs the two halves as separate graphs.
enum { PH_FRONT = 1, PH_BACK = 2, PH_ALL = 3 };
// search_nb / search_nbt > 0 (the search branch's callers): conv_search x search_nb (models/rpn.py:50-54, N-fused) is issued HERE, behind adjust
// and in front of the sequence's flush, so that it becomes the persistent launch's last record (round 6, smk_tune "seq_search"): no kernel
// boundary, no cold start, the weights behind the team's own L2 -- the register-fed 128 x 256 tiles run at 0.58 us per K tile inside the
// sequence where the stand-alone launch needs 0.9.  *search_done tells the caller that conv_search has been issued (recorded or launched).
static int run_backbone(smk_ctx *c, const float *x, int B, int S, hipStream_t s, int phase = PH_ALL, int search_nb = 0, int search_nbt = 0,
                        bool *search_done = nullptr) {
    // DT_F16X3: the trunk's tensors are stored as [hi | lo] planes -- twice the channels (smk_kernels.h)
    auto X = [c](int C) { return c->dtype == DT_F16X3 ? X3_PLANES * C : C; };
    const int s0 = (S - 7) / 2 + 1;          // conv1 7x7 s2 p0
    const int s1 = (s0 + 2 - 3) / 2 + 1;     // maxpool 3/2/1
    const int s2 = (s1 - 3) / 2 + 1;         // layer2 3x3 s2 p0
    Act p0 = act(c, "p0", s0, s0, X(64));
    Act x1 = act(c, "x1", s1, s1, X(64));
    auto stem_it = c->conv.find("stem");
    if (stem_it == c->conv.end()) return fail(SMK_E_STATE, "internal: conv stem not packed");
    if (!(phase & PH_FRONT)) {
        // the front half ran as its own graph: p1 is where it left it
    } else if (c->dtype == DT_F16 && g_tune.stem_fused && stem_it->second.w_frag) {
        // one launch: frame -> p0 (kept for Refine) -> pooled x1, the p0 tile never leaves LDS in between (stem_pool.hip)
        StemPoolParams sp_{x, stem_it->second.w_frag, stem_it->second.bias, p0.p, x1.p, B, S, s0, s1, stem_it->second.Kpad, c->wave_prio_now};
        ProfScope ps(c, s, "stem_pool", "stem_pool", 2.0 * B * s0 * s0 * 64.0 * 147.0,
                     (double)B * (3.0 * S * S * 4 + ((double)s0 * s0 + (double)s1 * s1) * 64 * 2));
        if (launch_stem_pool(sp_, s)) return fail(SMK_E_HIP, "stem_pool launch failed: %s", hipGetErrorString(hipGetLastError()));
    } else if (c->dtype == DT_F16X3) {
        // split operands: frame -> [hi | lo] x 8 channels, the 7x7 stem as a generic convolution on them, the pool on hi + lo
        Act xin = act(c, "xin", S, S, X3_PLANES * 8);
        CvtInParams ci{x, xin.p, B, 3, S, S, 8, 0, 1};
        {
            ProfScope ps(c, s, "cvt_in", "cvt_in_x3", 0.0, (double)B * S * S * (3 * 4 + X3_PLANES * 8 * 2));
            if (launch_cvt_in_x3(ci, s)) return fail(SMK_E_HIP, "cvt_in_x3 launch failed");
        }
        ConvOpt o;
        o.stride = 2; o.relu = 1;
        CHK(run_conv(c, "stem", xin, &p0, B, o, s));
        PoolParams pp{p0.p, x1.p, B, s0, s0, 64, s1, s1, 1};
        {
            ProfScope ps(c, s, "maxpool", "maxpool_x3", 0.0, (double)B * X3_PLANES * 64 * 2 * (s0 * s0 + s1 * s1));
            if (launch_maxpool_x3(pp, s)) return fail(SMK_E_HIP, "maxpool_x3 launch failed");
        }
    } else {
        Act xin = act(c, "xin", S, (S + 1) / 2, 8);            // pixel-pair layout (see pack_stem)
        CvtInParams ci{x, xin.p, B, 3, S, S, 8, 1};
        {
            ProfScope ps(c, s, "cvt_in", "cvt_in", 0.0, (double)B * S * S * (3 * 4 + 8 * esize(c->dtype)));
            if (launch_cvt_in(ci, kdtype(c->dtype), s)) return fail(SMK_E_HIP, "cvt_in launch failed");
        }
        ConvOpt o;
        o.stride = 2; o.stride_x = 1; o.relu = 1;
        CHK(run_conv(c, "stem", xin, &p0, B, o, s));
        PoolParams pp{p0.p, x1.p, B, s0, s0, 64, s1, s1};
        {
            ProfScope ps(c, s, "maxpool", "maxpool", 0.0, (double)B * 64 * esize(c->dtype) * (s0 * s0 + s1 * s1));
            if (launch_maxpool(pp, kdtype(c->dtype), s)) return fail(SMK_E_HIP, "maxpool launch failed");
        }
    }

    Act cur = x1;
    int sp = s1;                              // current spatial size
    // B >= 8, fp16: the three ResNet stages + adjust run as persistent per-XCD sequences (image b on XCD b % 8)
    struct SeqScope {
        smk_ctx *c;
        SeqScope(smk_ctx *c_, bool on) : c(c_) { c->seq_on = on; }
        ~SeqScope() { c->seq_on = false; c->seq_rec.clear(); c->seq_wstd.clear(); c->seq_ids.clear(); c->seq_flop = c->seq_bytes = 0.0; }
    } seq_scope(c, false);
    const bool seq_ok = seq_wanted(c, B) && !parallel_ok(c);
    bool c1_done = false;                     // the previous block's conv3 launch already computed this block's conv1 (run_conv_pair)
    bool adjust_done = false;
    for (int st = 0; st < 3; ++st) {
        // layer1 stays on the per-launch kernels: short K and 63 tiles of 64 rows per image (two rounds for 32 workgroups)
        // made it 140 us inside the sequence against 96 us as launches (SMK_SEQ_CLK, profiles/r02_seq_ab.txt)
        if (st == 0 && !(phase & PH_FRONT)) { cur = act(c, "p1", s1, s1, X(256)); continue; }
        if (st == 1 && !(phase & PH_BACK)) { c->last_B = B; c->last_S = S; return 0; }
        if (seq_ok && st == g_tune.seq_first_stage) c->seq_on = true;
        const int planes = STAGE_PLANES[st];
        for (int b = 0; b < STAGE_BLOCKS[st]; ++b) {
            char idb[32];
            snprintf(idb, sizeof(idb), "l%d.%d.", st + 1, b);
            const std::string id = idb;
            const int stride = (st == 1 && b == 0) ? 2 : 1;
            const int dil = (st == 2 && b > 0) ? 2 : 1;
            const int pad2 = dil > 1 ? dil : 2 - stride;
            const int so = stride == 2 ? s2 : sp;
            if (st == 0 && c->dtype == DT_F16 && g_tune.l1_fused && !c->seq_on && !c->prof_split_l1) {
                // layer1: the whole Bottleneck in one launch, weights in registers, intermediates in LDS (l1_block.hip)
                auto f1 = c->conv.find(id + "c1"), f2 = c->conv.find(id + "c2"), f3 = c->conv.find(id + "c3");
                auto fd = c->conv.find(id + "ds");
                if (f1 == c->conv.end() || f2 == c->conv.end() || f3 == c->conv.end() || (b == 0 && fd == c->conv.end()) ||
                    !f1->second.w_frag16 || !f2->second.w_frag16 || !f3->second.w_frag16 || (b == 0 && !fd->second.w_frag16))
                    return fail(SMK_E_STATE, "internal: layer1 block %d not packed for l1_block_kernel", b);
                const bool last1 = b == STAGE_BLOCKS[0] - 1;
                const char *on = last1 ? "p1" : ((b & 1) ? "b" : "a");
                Act out1 = act(c, on, sp, sp, X(256));
                L1BlockParams lp;
                memset(&lp, 0, sizeof(lp));
                lp.x = cur.p; lp.y = out1.p;
                lp.w1 = f1->second.w_frag16; lp.w2 = f2->second.w_frag16; lp.w3 = f3->second.w_frag16;
                lp.b1 = f1->second.bias; lp.b2 = f2->second.bias; lp.b3 = f3->second.bias;
                lp.K1pad = f1->second.Kpad; lp.K2pad = f2->second.Kpad; lp.K3pad = f3->second.Kpad;
                if (b == 0) { lp.wd = fd->second.w_frag16; lp.bd = fd->second.bias; lp.Kdpad = fd->second.Kpad; }
                lp.B = B; lp.S = sp; lp.Cin = cur.C; lp.prio = c->wave_prio_now;
                const double px = (double)B * sp * sp;
                const double flop = 2.0 * px * (64.0 * cur.C + 64.0 * 576 + 256.0 * 64 + (b == 0 ? 256.0 * 64 : 0.0));
                const double bytes = px * (cur.C + 256.0) * 2 + (64.0 * cur.C + 64 * 576 + 256 * 64 + (b == 0 ? 256 * 64 : 0)) * 2;
                const bool want_clk = getenv("SMK_L1_CLK") != nullptr && !c->graph_mode;
                if (want_clk && !c->seq_clk) HIPCHK(hipMalloc((void **)&c->seq_clk, sizeof(unsigned long long) * (2 * SEQ_MAX + 1)));
                lp.clk = want_clk ? c->seq_clk : nullptr;
                ProfScope ps(c, s, (id + "block").c_str(), "l1_block", flop, bytes);
                if (launch_l1_block(lp, s)) return fail(SMK_E_HIP, "l1_block launch failed: %s", hipGetErrorString(hipGetLastError()));
                if (want_clk) {
                    unsigned long long h[6];
                    HIPCHK(hipStreamSynchronize(s));
                    HIPCHK(hipMemcpy(h, c->seq_clk, sizeof(h), hipMemcpyDeviceToHost));
                    fprintf(stderr, "[l1 clk] %s workgroup 0: weights + halo in LDS %.2f | conv1 %.2f | conv2 %.2f | conv3 %.2f | stores %.2f | total %.2f us\n",
                            id.c_str(), (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0,
                            (h[5] - h[4]) / 100.0, (h[5] - h[0]) / 100.0);
                }
                cur = out1;
                continue;
            }
            // one layout per buffer (see build_arena): stage-private names, and layer2.0's pre-stride conv1 output on its own
            static const char *T1N[3] = {"t1", "t1_2", "t1_3"}, *T2N[3] = {"t2", "t2_2", "t2_3"}, *RN[3] = {"r", "r_2", "r_3"},
                              *AN[3] = {"a", "a_2", "a_3"}, *BN[3] = {"b", "b_2", "b_3"};
            Act t1 = act(c, stride == 2 ? "t1_s" : T1N[st], sp, sp, X(planes));
            Act t2 = act(c, T2N[st], so, so, X(planes));
            ConvOpt o1; o1.relu = 1;
            ConvOpt o2; o2.relu = 1; o2.stride = stride; o2.pad = pad2; o2.dil = dil;
            Act res = cur;
            const bool par = parallel_ok(c);
            const std::string id_ds = id + "ds", id_c1 = id + "c1";
            if (b == 0) {
                // the shortcut conv only depends on the block input: it shares a launch with conv1
                Act r = act(c, RN[st], so, so, X(planes * 4));
                ConvOpt od;
                if (st == 0) { od.stride = 1; od.pad = 0; }          // 1x1
                else if (st == 1) { od.stride = 2; od.pad = 0; }     // 3x3 s2 p0
                else { od.stride = 1; od.pad = 1; }                  // 3x3 s1 p1
                if (par) {
                    hipStream_t sd = c->side[0];
                    CHK(stream_dep(c, s, sd));
                    CHK(run_conv(c, id_ds.c_str(), cur, &r, B, od, sd));
                    CHK(run_conv(c, id_c1.c_str(), cur, &t1, B, o1, s));
                } else {
                    CHK(run_conv_jobs(c, {{id_ds.c_str(), &cur, &r, od}, {id_c1.c_str(), &cur, &t1, o1}}, B, 0, s));
                }
                res = r;
            } else if (!c1_done) {
                CHK(run_conv(c, id_c1.c_str(), cur, &t1, B, o1, s));
            }
            c1_done = false;
            CHK(run_conv(c, (id + "c2").c_str(), t1, &t2, B, o2, s));
            if (b == 0 && par) CHK(stream_dep(c, c->side[0], s));
            const bool last = b == STAGE_BLOCKS[st] - 1;
            const char *oname = last ? (st == 0 ? "p1" : st == 1 ? "p2" : AN[2]) : ((b & 1) ? BN[st] : AN[st]);
            if (last && st == 2 && cur.p == c->buf.at(AN[2])) oname = BN[2];
            Act out = act(c, oname, so, so, X(planes * 4));
            if (last && st == 2) c->p3_buf = oname;
            ConvOpt o3; o3.relu = 1; o3.res = &res; o3.res_mode = RES_PRE_RELU;
            // outside the persistent sequence: conv3 and the NEXT 1x1 convolution (the following block's conv1, or adjust behind
            // layer3 on the search branch) as one launch where the pair routine has the shape (layer2 / layer3 identity blocks)
            int paired = 1;
            if (!c->seq_on && st >= 1) {
                if (!last) {
                    char idn[32];
                    snprintf(idn, sizeof(idn), "l%d.%d.c1", st + 1, b + 1);
                    Act t1n = act(c, T1N[st], so, so, X(planes));
                    ConvOpt o1n; o1n.relu = 1;
                    paired = run_conv_pair(c, (id + "c3").c_str(), t2, out, o3, idn, t1n, o1n, B, s);
                    if (paired < 0) return paired;
                    if (paired == 0) c1_done = true;
                } else if (st == 2 && so >= 20) {
                    Act se = act(c, "search", so, so, X(256));
                    ConvOpt oa0;
                    paired = run_conv_pair(c, (id + "c3").c_str(), t2, out, o3, "adjust", se, oa0, B, s);
                    if (paired < 0) return paired;
                    if (paired == 0) adjust_done = true;
                }
            }
            if (paired == 1) CHK(run_conv(c, (id + "c3").c_str(), t2, &out, B, o3, s));
            cur = out;
            sp = so;
        }
    }
    // adjust: 1x1 1024->256 + BN, no ReLU; template (15 < 20): centre crop [4:-4] (custom.py:21-24)
    ConvOpt oa;
    if (sp < 20) {
        Act zf = act(c, "zf", sp - 8, sp - 8, X(256));
        oa.win = true; oa.Hl = oa.Wl = sp - 8; oa.org_y = oa.org_x = 4;
        CHK(run_conv(c, "adjust", cur, &zf, B, oa, s));
    } else if (!adjust_done) {
        Act se = act(c, "search", sp, sp, X(256));
        CHK(run_conv(c, "adjust", cur, &se, B, oa, s));
    }
    if (c->seq_on && search_nb > 0 && sp >= 20 && g_tune.seq_search && c->seq_rec.size() < (size_t)SEQ_MAX) {
        Act se = act(c, "search", sp, sp, X(256));
        Act xs = act(c, "xs", sp - 2, sp - 2, X(256 * search_nbt));
        ConvOpt o; o.relu = 1; o.n_override = 256 * search_nb;
        CHK(run_conv(c, "conv_search", se, &xs, B, o, s));         // (not eligible for the sequence: flushed + launched, still done)
        if (search_done) *search_done = true;
    }
    if (c->seq_on) CHK(seq_flush(c, B, s));
    c->last_B = B; c->last_S = S;
    return 0;
}

static int seq_template(smk_ctx *c, const float *z, int B, hipStream_t s) {
    auto X = [c](int C) { return c->dtype == DT_F16X3 ? X3_PLANES * C : C; };
    CHK(run_backbone(c, z, B, 127, s));
    const int nb = nbranch(c);
    Act zf = act(c, "zf", 7, 7, X(256));
    Act zk = act(c, "zk", 5, 5, X(256 * nb));
    ConvOpt o; o.relu = 1;                       // conv_kernel: 3x3 p0 + BN + ReLU (rpn.py:45-49), all branches fused
    CHK(run_conv(c, "conv_kernel", zf, &zk, B, o, s));
    return 0;
}

// defer_mask_join: the 63x63 mask head (HBM-write bound, nothing on the device reads it) is forked
// to a side stream and only joined by the caller at the end of the frame step, so that it runs
// beside the small decode / Refine launches instead of in front of them
static int seq_track(smk_ctx *c, const float *x, int B, int flags, float *cls, float *loc, float *mask,
                     hipStream_t s, bool defer_mask_join = false, int phase = PH_ALL) {
    auto X = [c](int C) { return c->dtype == DT_F16X3 ? X3_PLANES * C : C; };
    const bool x3 = c->dtype == DT_F16X3;
    const int nbt = nbranch(c);                                   // branches laid out in the buffers
    const int nb = (flags & SMK_TRACK_MASK) ? nbt : 2;            // branches computed
    bool search_done = false;
    CHK(run_backbone(c, x, B, 255, s, phase, nb, nbt, &search_done));
    Act se = act(c, "search", 31, 31, X(256));
    Act xs = act(c, "xs", 29, 29, X(256 * nbt));
    ConvOpt o; o.relu = 1; o.n_override = 256 * nb;               // conv_search x nb as one N-fused GEMM
    if (!search_done) CHK(run_conv(c, "conv_search", se, &xs, B, o, s));
    if (c->pipe_gate_late) {
        // pipelined step without the persistent sequence: nothing up to here writes what the previous frame's tail reads (p0 / p1 / p2
        // exist twice), so the gate sits HERE -- the tail has the whole backbone of this frame to finish beside -- and the heads below
        // (corr, head0, the decoded position) are the first writers it protects
        if (launch_pipe_gate(c->pipe_cnt, c->seq_err, c->seq_err_hdev, s, 0, nullptr, c->pipe_cnt + 8)) return fail(SMK_E_HIP, "pipe_gate launch failed");
        c->cap_has_seq = true;
    }
    Act corr = act(c, "corr", 25, 25, X(256 * nbt));
    Act h0 = act(c, "head0", 25, 25, X(256 * nbt));
    const bool par = parallel_ok(c);
    const bool want_mask = (flags & SMK_TRACK_MASK) && !(flags & SMK_TRACK_NO_MASK_HEAD);
    hipStream_t s_loc = par ? c->side[0] : s, s_cls = (par && want_mask) ? c->side[1] : s;
    // fp16: correlation + head.0 + cls / loc head.3 as ONE launch (corr_head.hip); the mask branch's head.3 follows as before
    bool fused_heads = false;
    if (c->dtype == DT_F16 && g_tune.corr_head && !par) {
        auto ih = c->conv.find("head0"), ic = c->conv.find("cls3"), il = c->conv.find("loc3");
        if (ih != c->conv.end() && ic != c->conv.end() && il != c->conv.end() && ih->second.w_frag && ic->second.w_frag && il->second.w_frag &&
            ih->second.Kpad == 256 && ic->second.Kpad == 256 && il->second.Kpad == 256 && ih->second.group_rows == 256 &&
            ic->second.N == 10 && il->second.N == 20) {
            CorrHeadParams hp;
            memset(&hp, 0, sizeof(hp));
            hp.xs = (const _Float16 *)xs.p; hp.zk = (const _Float16 *)c->buf.at("zk"); hp.corr = (_Float16 *)corr.p; hp.h0 = (_Float16 *)h0.p;
            hp.w0_frag = ih->second.w_frag; hp.b0 = ih->second.bias; hp.w0_bytes = (unsigned)((size_t)ih->second.rows * ih->second.Kpad * 2);
            hp.w3_frag[0] = ic->second.w_frag; hp.b3[0] = ic->second.bias; hp.out3[0] = cls; hp.n3[0] = 10;
            hp.w3_frag[1] = il->second.w_frag; hp.b3[1] = il->second.bias; hp.out3[1] = loc; hp.n3[1] = 20;
            hp.w3_bytes[0] = (unsigned)((size_t)ic->second.rows * ic->second.Kpad * 2);
            hp.w3_bytes[1] = (unsigned)((size_t)il->second.rows * il->second.Kpad * 2);
            hp.B = B; hp.nb = nb; hp.Cs = 256 * nbt;
            if (c->pipe_corr_sem) { hp.start_sem = c->pipe_cnt + 7; c->pipe_seq_exit_done = true; }
            const double flop = 2.0 * B * nb * 256.0 * 625 * 25 + 2.0 * B * nb * 625.0 * 256 * 256 + 2.0 * B * 625.0 * 256 * 30;
            const double bytes = (double)B * nb * 256.0 * (29 * 29 + 25 + 2 * 625) * 2 + 3.0 * 256 * 256 * 2 + (double)B * 30 * 625 * 4;
            ProfScope ps(c, s, "dw_xcorr+head0+cls3+loc3", "corr_head", flop, bytes);
            if (launch_corr_head(hp, s)) return fail(SMK_E_HIP, "corr_head launch failed: %s", hipGetErrorString(hipGetLastError()));
            fused_heads = true;
        }
    }
    if (!fused_heads) {
    XcorrParams xp{xs.p, c->buf.at("zk"), corr.p, B, 29, 29, 5, 5, 25, 25, 256 * nb, 256 * nbt};
    {
        // algorithmic bytes per branch-item: read 256*(29*29 + 5*5), write 256*25*25 elements
        const double xb = (double)B * nb * 256.0 * (29 * 29 + 25 + 625) * esize(c->dtype);
        ProfScope ps(c, s, "dw_xcorr", x3 ? "dw_xcorr_x3" : "dw_xcorr", 2.0 * B * nb * 256.0 * 625 * 25, x3 ? (double)X3_PLANES * xb : xb);
        // (split operands: xs / zk in whole-tensor planes, corr in per-branch planes -- head.0 is a grouped convolution)
        if (x3 ? launch_xcorr_x3(xp, s) : launch_xcorr(xp, kdtype(c->dtype), s)) return fail(SMK_E_HIP, "xcorr launch failed");
    }
    ConvOpt oh; oh.relu = 1; oh.groups = nb;                      // head.0 1x1 + BN + ReLU per branch
    CHK(run_conv(c, "head0", corr, &h0, B, oh, s));
    // the three head.3 convs are independent: cls / loc / mask side by side
    if (par) { CHK(stream_dep(c, s, s_loc)); if (s_cls != s) CHK(stream_dep(c, s, s_cls)); }
    ConvOpt oc; oc.nchw_out = cls; oc.cin_off = 0;
    ConvOpt ol; ol.nchw_out = loc; ol.cin_off = x3 ? X3_PLANES * 256 : 256;      // (x3: a branch's [hi | lo] planes side by side)
    if (par) {
        CHK(run_conv(c, "cls3", h0, nullptr, B, oc, s_cls));
        CHK(run_conv(c, "loc3", h0, nullptr, B, ol, s_loc));
    } else {
        CHK(run_conv_jobs(c, {{"cls3", &h0, nullptr, oc}, {"loc3", &h0, nullptr, ol}}, B, 1, s));
    }
    }
    if (want_mask) {
        ConvOpt om; om.nchw_out = mask; om.cin_off = x3 ? 2 * X3_PLANES * 256 : 512;      // (x3: plain fp16 pack on the hi plane of the mask branch)
        if (c->defer_mask_req && !par) {
            // handed to seq_refine: it runs inside the chain launch, beside the (B-workgroup) Refine chain
            auto it = c->conv.find("mask3");
            if (it == c->conv.end()) return fail(SMK_E_STATE, "internal: conv mask3 not packed");
            CHK(conv_params(c, it->second, h0, nullptr, B, om, c->deferred_mask));
            const ConvParams &mp = c->deferred_mask;
            c->deferred_mask_flop = 2.0 * mp.M * (double)mp.N * mp.kh * mp.kw * mp.Ci;
            c->deferred_mask_bytes = (double)mp.M * mp.Ci * esize(c->dtype) + (double)mp.M * mp.N * 4 + (double)mp.N * mp.Ci * esize(c->dtype);
            c->have_deferred_mask = true;
        } else if (defer_mask_join && !par && !c->prof && g_tune.mask_overlap && c->side[0]) {
            CHK(stream_dep(c, s, c->side[0]));
            CHK(run_conv(c, "mask3", h0, nullptr, B, om, c->side[0]));
            c->mask_join_pending = true;
        } else {
            CHK(run_conv(c, "mask3", h0, nullptr, B, om, s));
        }
    }
    if (par) { CHK(stream_dep(c, s_loc, s)); if (s_cls != s) CHK(stream_dep(c, s_cls, s)); }
    c->last_nb = nb;
    return 0;
}

// Refine.forward(test=True), custom.py:131-154, with a per-item position
// part: 0 = all of it; 1 = the window convolutions + deconv + v*.2 (what reads the kept features and the position); 2 = the chain launch
// (what reads only part 1's outputs and, for the mask head, head0).  The split exists for the fp16 chain path only (depth-2 pipelining).
static bool refine_splittable(const smk_ctx *c, int B) {
    return kdtype(c->dtype) == DT_F16 && g_tune.chain && !parallel_ok(c) && g_tune.merge && (g_tune.merge == 2 || B <= g_tune.merge_max_batch);
}
static int seq_refine(smk_ctx *c, int B, float *out, hipStream_t s, int part = 0) {
    const int *pos = c->pos_dev;
    // (DT_F16X3: Refine runs in plain fp16 on the hi planes of the kept trunk tensors -- channel stride 2 C, first C channels)
    auto X = [c](int C) { return c->dtype == DT_F16X3 ? X3_PLANES * C : C; };
    Act corr = act(c, "corr", 25, 25, X(256 * 3));
    Act p0 = act(c, "p0", 125, 125, X(64)), p1 = act(c, "p1", 63, 63, X(256)), p2 = act(c, "p2", 31, 31, X(512));
    // The three window convs v2.0 / v1.0 / v0.0 (the heavy part of Refine) depend only on the
    // kept backbone features and pos: they run on a side stream beside deconv -> h2 -> ...
    ConvOpt r3; r3.pad = 1; r3.relu = 1;
    ConvOpt w2 = r3; w2.win = true; w2.Hl = w2.Wl = 15; w2.pos = pos; w2.pos_mul = 1; w2.pos_add = -4;   // pad 4 (:135)
    ConvOpt w1 = r3; w1.win = true; w1.Hl = w1.Wl = 31; w1.pos = pos; w1.pos_mul = 2; w1.pos_add = -8;   // pad 8 (:134)
    ConvOpt w0 = r3; w0.win = true; w0.Hl = w0.Wl = 61; w0.pos = pos; w0.pos_mul = 4; w0.pos_add = -16;  // pad 16 (:133)
    const bool par = parallel_ok(c);
    hipEvent_t ev_v2 = nullptr, ev_v1 = nullptr, ev_v0 = nullptr;
    if (par) {
        hipStream_t sd = c->side[0];
        CHK(stream_dep(c, s, sd));
        Act v2a_ = act(c, "rf_v2a", 15, 15, 128), v1a_ = act(c, "rf_v1a", 31, 31, 64), v0a_ = act(c, "rf_v0a", 61, 61, 16);
        CHK(run_conv(c, "v2.0", p2, &v2a_, B, w2, sd));
        ev_v2 = c->ev_pool[c->ev_next++ % c->ev_pool.size()];
        HIPCHK(hipEventRecord(ev_v2, sd));
        CHK(run_conv(c, "v1.0", p1, &v1a_, B, w1, sd));
        ev_v1 = c->ev_pool[c->ev_next++ % c->ev_pool.size()];
        HIPCHK(hipEventRecord(ev_v1, sd));
        CHK(run_conv(c, "v0.0", p0, &v0a_, B, w0, sd));
        ev_v0 = c->ev_pool[c->ev_next++ % c->ev_pool.size()];
        HIPCHK(hipEventRecord(ev_v0, sd));
    }
    // deconv(corr_feature[:, :, y, x]) -> [15,15,32]            (:145,:149)
    Act d1 = act(c, "rf_d", 1, 1, 15 * 15 * 32);
    ConvOpt od; od.win = true; od.Hl = od.Wl = 1; od.pos = pos; od.pos_mul = 1; od.cin_off = c->dtype == DT_F16X3 ? 2 * X3_PLANES * 256 : 512;
    Act v2a = act(c, "rf_v2a", 15, 15, 128), v1a = act(c, "rf_v1a", 31, 31, 64), v0a = act(c, "rf_v0a", 61, 61, 16);
    const bool merged = !par && (!c->prof || c->prof_merge) && g_tune.merge && (g_tune.merge == 2 || B <= g_tune.merge_max_batch);
    if (merged && part != 2) {
        // the window convs only depend on the kept backbone features and pos: one launch with deconv
        w2.tile_code = 4;     // 64x64 (256-byte K tile): v2.0's long K chain sets the pace
        // smk_tune "rf_wreg": bit 0 the four members on the register-fed kernel, bits 4..7 its tile code (0 = 3: 64x64; 6 = 128x64; 8 = 32x64)
        if (g_tune.rf_wreg & 1) w2.wreg = w1.wreg = w0.wreg = od.wreg = ((g_tune.rf_wreg >> 4) & 15) ? ((g_tune.rf_wreg >> 4) & 15) : 3;
        CHK(run_conv_jobs(c, {{"v2.0", &p2, &v2a, w2}, {"v1.0", &p1, &v1a, w1}, {"v0.0", &p0, &v0a, w0},
                              {"deconv", &corr, &d1, od}}, B, 0, s));
    } else if (part != 2) {
        CHK(run_conv(c, "deconv", corr, &d1, B, od, s));
    }
    Act d = act(c, "rf_d", 15, 15, 32);
    if (kdtype(c->dtype) == DT_F16 && g_tune.chain && !par) {       // (split-operand contexts: Refine is plain fp16 on the hi planes -- same launches)
        // fp16: v*.2 in one merged launch (they only depend on v*.0), then the nine sequential convolutions
        // h2 -> post0 -> h1 -> post1 -> h0 -> post2 as ONE launch with the activations in LDS (refine_chain.hip)
        if (!merged && part != 2) {
            CHK(run_conv(c, "v2.0", p2, &v2a, B, w2, s));
            CHK(run_conv(c, "v1.0", p1, &v1a, B, w1, s));
            CHK(run_conv(c, "v0.0", p0, &v0a, B, w0, s));
        }
        Act V2 = act(c, "rf_s2", 15, 15, 32), V1 = act(c, "rf_s1", 31, 31, 16), V0 = act(c, "rf_s0", 61, 61, 8);
        ConvOpt r3l = r3;
        r3l.tile_code = g_tune.rf_tile2;          // (A/B knob: workgroup tile of the merged v*.2 launch; 0 = the lead's own choice, 64x64)
        ConvOpt r3w = r3;
        if ((g_tune.rf_wreg & 2) && merged) r3l.wreg = r3w.wreg = ((g_tune.rf_wreg >> 8) & 15) ? ((g_tune.rf_wreg >> 8) & 15) : 3;      // bit 1: the v*.2 launch, bits 8..11 its tile code
        if (part != 2) CHK(run_conv_jobs(c, {{"v2.2", &v2a, &V2, r3l}, {"v1.2", &v1a, &V1, r3w}, {"v0.2", &v0a, &V0, r3w}}, B, 0, s));
        if (part == 1) return 0;
        static const char *ids[9] = {"h2.0", "h2.2", "post0", "h1.0", "h1.2", "post1", "h0.0", "h0.2", "post2"};
        static const int geo[9][3] = {{225, 32, 32}, {225, 32, 32}, {961, 32, 16}, {961, 16, 16}, {961, 16, 16},
                                      {3721, 16, 4}, {3721, 4, 4}, {3721, 4, 4}, {16129, 4, 1}};   // pixels, Cin, Cout
        RefineChainParams rp;
        double flop = 0.0, wbytes = 0.0;
        for (int i = 0; i < 9; ++i) {
            auto it = c->conv.find(ids[i]);
            if (it == c->conv.end()) return fail(SMK_E_STATE, "internal: conv %s not packed", ids[i]);
            const PackedConv &pc = it->second;
            if (pc.k != 3 || pc.groups != 1 || pc.Ci < geo[i][1] || pc.N != geo[i][2])
                return fail(SMK_E_STATE, "internal: conv %s does not have the Refine geometry", ids[i]);
            rp.L[i] = RefineChainLayer{pc.w, pc.bias, pc.Kpad, pc.Ci};
            flop += 2.0 * B * geo[i][0] * 9.0 * geo[i][1] * geo[i][2];
            wbytes += 2.0 * 9.0 * geo[i][1] * geo[i][2];
        }
        rp.d = d.p;
        rp.v2 = V2.p; rp.v1 = V1.p; rp.v0 = V0.p;
        rp.v2_cs = V2.C; rp.v1_cs = V1.C; rp.v0_cs = V0.C;
        rp.out = out;
        rp.B = B;
        rp.clk = nullptr;
        rp.ring = nullptr; rp.ring_cursor = nullptr; rp.ring_done = nullptr; rp.ring_rows = 0;
        rp.tail_sem = nullptr;
        if (c->ring_in_step && c->ring_ref) {            // the frame's fp16 logits go to the result ring from post2 itself
            rp.ring = (_Float16 *)c->ring_ref; rp.ring_cursor = c->ring_cursor; rp.ring_done = (unsigned *)(c->ring_cursor + 1);
            rp.ring_rows = c->ring_rows;
            c->ring_ref_folded = true;
        }
        // SMK_CHAIN_CLK=1 (eager runs only): print the time workgroup 0 spends in each of the nine layers
        static const bool want_clk = getenv("SMK_CHAIN_CLK") != nullptr;
        static unsigned long long *clk_dev = nullptr;
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        if (want_clk && hipStreamIsCapturing(s, &cst) == hipSuccess && cst == hipStreamCaptureStatusNone) {
            if (!clk_dev) HIPCHK(hipMalloc((void **)&clk_dev, 32 * sizeof(unsigned long long)));
            rp.clk = clk_dev;
        }
        const double cbytes = B * (2.0 * (7200 + 225 * 32 + 961 * 16 + 3721 * 4) + 4.0 * 16129) + wbytes;
        if (c->have_deferred_mask && !rp.clk) {
            ConvBatch cb;
            cb.n = 1;
            cb.p[0] = c->deferred_mask;
            ProfScope ps(c, s, "refine_chain+mask3", "chain_mask", flop + c->deferred_mask_flop, cbytes + c->deferred_mask_bytes);
            if (c->pipe_tail_fold) rp.tail_sem = c->pipe_cnt;      // pipelined step: this launch ends the tail, its last workgroup is the "done" mark
            const int rc = launch_chain_mask(rp, cb, s);
            rp.tail_sem = nullptr;
            if (rc == 0) {
                c->have_deferred_mask = false;
                if (c->pipe_tail_fold) c->pipe_done_folded = true;
                return 0;
            }
            if (rc != 1) return fail(SMK_E_HIP, "chain_mask launch failed: %s", hipGetErrorString(hipGetLastError()));
            ps.cancel();      // (odd tile count: the mask head keeps its own launch -- smk_step runs it after the chain)
        }
        ProfScope ps(c, s, "refine_chain", "refine_chain", flop, cbytes);
        if (launch_refine_chain(rp, s)) return fail(SMK_E_HIP, "refine_chain launch failed: %s", hipGetErrorString(hipGetLastError()));
        if (rp.clk) {
            unsigned long long hh[22];
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(hh, clk_dev, sizeof(hh), hipMemcpyDeviceToHost));
            {
                const unsigned long long *h = hh;
                fprintf(stderr, "refine_chain layers (us): load=%.2f", (double)(h[1] - h[0]) * 0.01);
                for (int i = 0; i < 9; ++i) fprintf(stderr, " %s=%.2f", ids[i], (double)(h[i + 2] - h[i + 1]) * 0.01);
                fprintf(stderr, " total=%.2f\n", (double)(h[10] - h[0]) * 0.01);
            }
        }
        return 0;
    }
    // stage 2 @15x15                                             (:150)
    Act h2a = act(c, "rf_h2a", 15, 15, 32), h2b = act(c, "rf_h2b", 15, 15, 32);
    CHK(run_conv(c, "h2.0", d, &h2a, B, r3, s));
    CHK(run_conv(c, "h2.2", h2a, &h2b, B, r3, s));
    Act s2 = act(c, "rf_s2", 15, 15, 32);
    if (par) CHK(hipStreamWaitEvent(s, ev_v2, 0) == hipSuccess ? 0 : fail(SMK_E_HIP, "wait ev_v2"));
    else if (!merged) CHK(run_conv(c, "v2.0", p2, &v2a, B, w2, s));
    ConvOpt a2 = r3; a2.res = &h2b; a2.res_mode = RES_POST_RELU;
    CHK(run_conv(c, "v2.2", v2a, &s2, B, a2, s));
    Act u0 = act(c, "rf_u0", 31, 31, 16);
    ConvOpt pu0; pu0.pad = 1; pu0.ups = true; pu0.Hl = pu0.Wl = 31;
    CHK(run_conv(c, "post0", s2, &u0, B, pu0, s));
    // stage 1 @31x31                                             (:151)
    Act h1a = act(c, "rf_h1a", 31, 31, 16), h1b = act(c, "rf_h1b", 31, 31, 16);
    CHK(run_conv(c, "h1.0", u0, &h1a, B, r3, s));
    CHK(run_conv(c, "h1.2", h1a, &h1b, B, r3, s));
    Act s1 = act(c, "rf_s1", 31, 31, 16);
    if (par) CHK(hipStreamWaitEvent(s, ev_v1, 0) == hipSuccess ? 0 : fail(SMK_E_HIP, "wait ev_v1"));
    else if (!merged) CHK(run_conv(c, "v1.0", p1, &v1a, B, w1, s));
    ConvOpt a1 = r3; a1.res = &h1b; a1.res_mode = RES_POST_RELU;
    CHK(run_conv(c, "v1.2", v1a, &s1, B, a1, s));
    Act u1 = act(c, "rf_u1", 61, 61, 8);
    ConvOpt pu1; pu1.pad = 1; pu1.ups = true; pu1.Hl = pu1.Wl = 61;
    CHK(run_conv(c, "post1", s1, &u1, B, pu1, s));
    // stage 0 @61x61                                             (:152)
    Act h0a = act(c, "rf_h0a", 61, 61, 8), h0b = act(c, "rf_h0b", 61, 61, 8);
    CHK(run_conv(c, "h0.0", u1, &h0a, B, r3, s));
    CHK(run_conv(c, "h0.2", h0a, &h0b, B, r3, s));
    Act s0 = act(c, "rf_s0", 61, 61, 8);
    if (par) CHK(hipStreamWaitEvent(s, ev_v0, 0) == hipSuccess ? 0 : fail(SMK_E_HIP, "wait ev_v0"));
    else if (!merged) CHK(run_conv(c, "v0.0", p0, &v0a, B, w0, s));
    ConvOpt a0 = r3; a0.res = &h0b; a0.res_mode = RES_POST_RELU;
    CHK(run_conv(c, "v0.2", v0a, &s0, B, a0, s));
    ConvOpt pu2; pu2.pad = 1; pu2.ups = true; pu2.Hl = pu2.Wl = 127; pu2.nchw_out = out;
    CHK(run_conv(c, "post2", s0, nullptr, B, pu2, s));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// graph capture / replay
// ---------------------------------------------------------------------------------------------
// capture `body` into an instantiated graph under `key` (LRU-bounded cache); no launch
template <typename F>
static int capture_graph(smk_ctx *c, const GraphKey &key, F &&body) {
    if (!c->cap_stream) HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
    const bool pend0 = c->seq_pending;
    c->cap_has_seq = false;
    int rc = body(c->cap_stream);
    const bool has_seq = c->cap_has_seq;
    c->seq_pending = pend0;                          // captured, not enqueued
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(c->cap_stream, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(SMK_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return fail(SMK_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    // bound the cache: drop the least recently used graph (a caller that hands over fresh I/O buffers every frame
    // re-captures every frame anyway -- that is what the staging path is for -- but it must not evict the graphs
    // of callers with stable buffers)
    while (c->graphs.size() >= 64) {
        auto lru = c->graph_used.begin();
        for (auto u = c->graph_used.begin(); u != c->graph_used.end(); ++u)
            if (u->second < lru->second) lru = u;
        auto g = c->graphs.find(lru->first);
        if (g != c->graphs.end()) { hipGraphExecDestroy(g->second); c->graphs.erase(g); }
        c->graph_has_seq.erase(lru->first);
        c->graph_used.erase(lru);
    }
    c->graphs.emplace(key, ex);
    c->graph_has_seq[key] = has_seq;
    c->graph_used[key] = ++c->graph_tick;
    return 0;
}

static int launch_graph(smk_ctx *c, const GraphKey &key, hipStream_t s) {
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) return fail(SMK_E_STATE, "internal: graph not captured");
    c->graph_used[key] = ++c->graph_tick;
    HIPCHK(hipGraphLaunch(it->second, s));
    if (c->graph_has_seq[key]) c->seq_pending = true;
    return 0;
}

template <typename F>
static int run_maybe_graph(smk_ctx *c, const GraphKey &key, hipStream_t s, F &&body) {
    if (!c->graph_mode || c->prof) return body(s);
    if (c->graphs.find(key) == c->graphs.end()) CHK(capture_graph(c, key, body));
    return launch_graph(c, key, s);
}

static void drop_graph(smk_ctx *c, const GraphKey &key) {
    auto g = c->graphs.find(key);
    if (g != c->graphs.end()) { hipGraphExecDestroy(g->second); c->graphs.erase(g); }
    c->graph_has_seq.erase(key);
    c->graph_used.erase(key);
}

// the pipelined step's semaphores at rest (device idle): one tail "completed" (the first frame has nothing to wait for), no main part
static int pipe_reset_counters(smk_ctx *c) {
    if (!c->pipe_cnt) return 0;
    unsigned init[16] = {1u, 0u};
    HIPCHK(hipMemcpy(c->pipe_cnt, init, sizeof(init), hipMemcpyHostToDevice));
    if (c->pipe_sig) { HIPCHK(hipMemset(c->pipe_sig, 0, 8)); c->pipe_sig_n = 0; }
    return 0;
}

static int launch_graph(smk_ctx *c, const GraphKey &key, hipStream_t s);
// depth-2 pipelining: the second part of the last frame's tail waits for a next frame that may never come -- whoever needs the
// results (or is about to drop the graphs) launches it without its gate
static int pipe_flush(smk_ctx *c) {
    if (!c->tail2_pending) return 0;
    c->tail2_pending = false;
    CHK(launch_graph(c, c->tail2_key, c->pipe_stream));
    hipEvent_t e = c->pipe_ev[c->pipe_ev_next++ % c->pipe_ev.size()];
    HIPCHK(hipEventRecord(e, c->pipe_stream));
    c->tail_ev = e;
    c->tail_pending = true;
    return 0;
}

// before the captured graphs are dropped: nothing of a pipelined step may still be waiting to be launched or running
static int pipe_quiesce(smk_ctx *c) {
    CHK(pipe_flush(c));
    if (c->pipe_stream) HIPCHK(hipStreamSynchronize(c->pipe_stream));
    c->tail_pending = false;
    return 0;
}

// order `s` behind the Refine / mask tail a pipelined smk_step left on the side stream (no-op when there is none)
static int pipe_join(smk_ctx *c, hipStream_t s, bool clear) {
    CHK(pipe_flush(c));
    if (!c->tail_pending) return 0;
    HIPCHK(hipStreamWaitEvent(s, c->tail_ev, 0));
    if (clear) c->tail_pending = false;
    return 0;
}

// conv_seq_kernel assumes that a one-block-per-CU launch puts the same number of blocks on every XCD (the dispatcher
// deals consecutive blocks round-robin over the XCDs -- observed, not a HIP guarantee) and that one workgroup of it
// (139 KB of LDS, 512 threads) is resident per CU: checked once per context with a census launch and the runtime's
// occupancy answer for THAT kernel; if either fails the per-launch kernels are used instead.  Returns the grid or 0.
static int seq_grid_for(int ncu) {
    if (ncu < 8 || ncu % 8 != 0 || ncu > 1024) return 0;
    std::vector<int> x(ncu, -1);
    const int crc = xcc_census(ncu, x.data());
    bool ok = crc == 0;
    int per[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; ok && i < ncu; ++i) {
        if (x[i] < 0 || x[i] > 7) ok = false;
        else per[x[i]]++;
    }
    for (int q = 0; ok && q < 8; ++q) ok = per[q] == ncu / 8;
    const int occ = conv_seq_occupancy();
    if (ok && occ < 1) ok = false;
    if (!ok && getenv("SMK_DEBUG"))
        fprintf(stderr, "[siammask_hip] XCD placement check failed (rc %d, %d CUs, per XCD %d %d %d %d %d %d %d %d, occupancy %d): "
                "persistent sequences off\n", crc, ncu, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], occ);
    return ok ? ncu : 0;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int smk_version(void) { return (1 << 16) | 6; }   // 1.6: SMK_DTYPE_F16X3 (split-operand fp16 contexts)
//   // 1.2: smk_decode / smk_step take float64 target_wh and write a float64 box; 1.3: smk_op_conv_seq,
                                                  // sequence failures reported at the next entry point

const char *smk_last_error(void) { return g_err.c_str(); }

int smk_create(smk_ctx **out, int device, int dtype, int variant, int max_batch) {
    if (!out) return fail(SMK_E_ARG, "smk_create: out is NULL");
    *out = nullptr;
    if (dtype != SMK_DTYPE_F32 && dtype != SMK_DTYPE_F16 && dtype != SMK_DTYPE_F16X3) return fail(SMK_E_ARG, "smk_create: bad dtype %d", dtype);
    if (variant < SMK_VARIANT_RPN || variant > SMK_VARIANT_SHARP) return fail(SMK_E_ARG, "smk_create: bad variant %d", variant);
    if (max_batch < 1 || max_batch > 1024) return fail(SMK_E_ARG, "smk_create: max_batch %d out of range", max_batch);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SMK_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(SMK_E_ARG, "smk_create: device %d of %d", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SMK_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    std::unique_ptr<smk_ctx> c(new smk_ctx);
    c->device = device; c->dtype = dtype; c->variant = variant; c->maxB = max_batch;
    const char *g = getenv("SMK_GRAPH");
    c->graph_mode = g && strcmp(g, "0") != 0;
    int rc = build_arena(c.get());
    if (rc) return rc;
    if (!zero_page()) return fail(SMK_E_HIP, "could not allocate the zero page");   // before any capture
    c->seq_grid = seq_grid_for(prop.multiProcessorCount);
    xcorr_prepare();
    for (int i = 0; i < 2; ++i) HIPCHK(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
    c->ev_pool.resize(64);
    for (auto &e : c->ev_pool) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    {
        const char *cc = getenv("SMK_CONCURRENCY");
        c->concurrency = !(cc && !strcmp(cc, "0")) && g_concurrency_default != 0;
        double w[625], h[25];
        for (int i = 0; i < 25; ++i) h[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / 24.0);   // np.hanning(25)
        for (int y = 0; y < 25; ++y)
            for (int x = 0; x < 25; ++x) w[y * 25 + x] = h[y] * h[x];                      // np.outer
        HIPCHK(hipMalloc((void **)&c->window_dev, sizeof(w)));
        HIPCHK(hipMemcpy(c->window_dev, w, sizeof(w), hipMemcpyHostToDevice));
    }
    *out = c.release();
    return 0;
}

int smk_destroy(smk_ctx *c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    if (c->pipe_stream) hipStreamSynchronize(c->pipe_stream);   // a tail may still read the arena
    for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second);
    if (c->cap_stream) hipStreamDestroy(c->cap_stream);
    for (auto &kv : c->buf)
        if (!c->buf_alias.count(kv.first)) hipFree(kv.second);
    for (auto &kv : c->conv) { hipFree(kv.second.w); hipFree(kv.second.w_halo); hipFree(kv.second.w_frag); hipFree(kv.second.w_frag16); hipFree(kv.second.w_frag_halo); hipFree(kv.second.bias); hipFree(kv.second.oscale); }
    if (c->pos_dev) hipFree(c->pos_dev);
    if (c->dec_scratch) hipFree(c->dec_scratch);
    if (c->ks_part) hipFree(c->ks_part);
    if (c->ks_cnt) hipFree(c->ks_cnt);
    if (c->seq_bar) hipFree(c->seq_bar);
    if (c->seq_xch) hipFree(c->seq_xch);
    if (c->seq_err) hipFree(c->seq_err);
    if (c->seq_err_host) hipHostFree(c->seq_err_host);
    if (c->seq_clk) hipFree(c->seq_clk);
    if (c->seq_clk2) hipFree(c->seq_clk2);
    if (c->window_dev) hipFree(c->window_dev);
    if (c->ring_cursor) hipFree(c->ring_cursor);
    for (auto &e : c->ev_pool) hipEventDestroy(e);
    for (auto &e : c->prof_pool) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (c->side[i]) hipStreamDestroy(c->side[i]);
    for (auto &e : c->pipe_ev) hipEventDestroy(e);
    if (c->pipe_stream) hipStreamDestroy(c->pipe_stream);
    if (c->pipe_cnt) hipFree(c->pipe_cnt);
    if (c->pipe_sig) hipFree(c->pipe_sig);
    delete c;
    return 0;
}

int smk_set_weight(smk_ctx *c, const char *name, const float *data, const int64_t *shape, int ndim) {
    if (!c || !name || (!data && ndim > 0)) return fail(SMK_E_ARG, "smk_set_weight: null argument");
    if (ndim < 0 || ndim > 4) return fail(SMK_E_ARG, "smk_set_weight(%s): ndim %d", name, ndim);
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return fail(SMK_E_ARG, "smk_set_weight(%s): negative dim", name);
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(data, data + n);
    c->host_w[name] = std::move(t);
    c->finalized = false;
    return 0;
}

int smk_finalize_weights(smk_ctx *c) {
    if (!c) return fail(SMK_E_ARG, "smk_finalize_weights: ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    CHK(pipe_quiesce(c));
    for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    c->graph_used.clear();
    for (auto &kv : c->conv) { hipFree(kv.second.w); hipFree(kv.second.w_halo); hipFree(kv.second.w_frag); hipFree(kv.second.w_frag16); hipFree(kv.second.w_frag_halo); hipFree(kv.second.bias); hipFree(kv.second.oscale); }
    c->conv.clear();
    int rc = build_weights(c);
    if (rc) return rc;
    c->finalized = true;
    c->template_B = 0;
    c->track_B = 0;
    return 0;
}

// ---- packed-weight cache: the BN-folded, MFMA-packed weights as one blob ----
// layout: PackHeader | per conv: PackEntry, weight bytes [rows][Kpad] (dtype), bias [rows] f32
struct PackHeader {
    char magic[8];              // "SMKPACK1"
    int32_t abi, dtype, variant, n_conv, npad_align, kpad_align;
    uint64_t total_bytes;
};
struct PackEntry {
    char id[24];
    int32_t N, rows, group_rows, groups, Ci, k, K, Kpad, kw, alg_k;
};

static size_t packed_bytes(const smk_ctx *c) {
    size_t n = sizeof(PackHeader);
    for (auto &kv : c->conv)
        n += sizeof(PackEntry) + (size_t)kv.second.rows * kv.second.Kpad * esize(c->dtype) + (size_t)kv.second.rows * 4;
    return n;
}

int smk_packed_size(smk_ctx *c, uint64_t *bytes) {
    if (!c || !bytes) return fail(SMK_E_ARG, "smk_packed_size: null argument");
    if (!c->finalized) return fail(SMK_E_STATE, "smk_packed_size: weights not finalized");
    *bytes = packed_bytes(c);
    return 0;
}

int smk_export_packed(smk_ctx *c, void *host_buf, uint64_t capacity) {
    if (!c || !host_buf) return fail(SMK_E_ARG, "smk_export_packed: null argument");
    if (!c->finalized) return fail(SMK_E_STATE, "smk_export_packed: weights not finalized");
    if (c->dtype == DT_F16X3) return fail(SMK_E_ARG, "smk_export_packed: split-operand contexts re-pack from the state dict (no blob format for the tripled K)");
    const size_t need = packed_bytes(c);
    if (capacity < need) return fail(SMK_E_ARG, "smk_export_packed: buffer too small (%zu needed)", need);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned char *o = (unsigned char *)host_buf;
    PackHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SMKPACK1", 8);
    h.abi = smk_version(); h.dtype = c->dtype; h.variant = c->variant; h.n_conv = (int)c->conv.size();
    h.npad_align = NPAD_ALIGN; h.kpad_align = KPAD_ALIGN; h.total_bytes = need;
    memcpy(o, &h, sizeof(h)); o += sizeof(h);
    for (auto &kv : c->conv) {
        const PackedConv &pc = kv.second;
        PackEntry en;
        memset(&en, 0, sizeof(en));
        if (kv.first.size() >= sizeof(en.id)) return fail(SMK_E_STATE, "internal: conv id too long");
        memcpy(en.id, kv.first.c_str(), kv.first.size());
        en.N = pc.N; en.rows = pc.rows; en.group_rows = pc.group_rows; en.groups = pc.groups;
        en.Ci = pc.Ci; en.k = pc.k; en.K = pc.K; en.Kpad = pc.Kpad; en.kw = pc.kw; en.alg_k = pc.alg_k;
        memcpy(o, &en, sizeof(en)); o += sizeof(en);
        const size_t wb = (size_t)pc.rows * pc.Kpad * esize(c->dtype), bb = (size_t)pc.rows * 4;
        HIPCHK(hipMemcpy(o, pc.w, wb, hipMemcpyDeviceToHost)); o += wb;
        HIPCHK(hipMemcpy(o, pc.bias, bb, hipMemcpyDeviceToHost)); o += bb;
    }
    return 0;
}

int smk_import_packed(smk_ctx *c, const void *host_buf, uint64_t bytes) {
    if (!c || !host_buf) return fail(SMK_E_ARG, "smk_import_packed: null argument");
    if (bytes < sizeof(PackHeader)) return fail(SMK_E_WEIGHT, "smk_import_packed: truncated blob");
    if (c->dtype == DT_F16X3) return fail(SMK_E_ARG, "smk_import_packed: split-operand contexts re-pack from the state dict");
    const unsigned char *p = (const unsigned char *)host_buf, *end = p + bytes;
    PackHeader h;
    memcpy(&h, p, sizeof(h)); p += sizeof(h);
    if (memcmp(h.magic, "SMKPACK1", 8) != 0) return fail(SMK_E_WEIGHT, "smk_import_packed: bad magic");
    if (h.abi != smk_version() || h.dtype != c->dtype || h.variant != c->variant || h.npad_align != NPAD_ALIGN ||
        h.kpad_align != KPAD_ALIGN)
        return fail(SMK_E_WEIGHT, "smk_import_packed: blob was packed for abi %#x dtype %d variant %d (ctx: %#x %d %d)",
                    h.abi, h.dtype, h.variant, smk_version(), c->dtype, c->variant);
    if (h.total_bytes != bytes || h.n_conv < 1 || h.n_conv > 256) return fail(SMK_E_WEIGHT, "smk_import_packed: size mismatch");
    HIPCHK(hipSetDevice(c->device));
    CHK(pipe_quiesce(c));
    for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    c->graph_used.clear();
    for (auto &kv : c->conv) { hipFree(kv.second.w); hipFree(kv.second.w_halo); hipFree(kv.second.w_frag); hipFree(kv.second.w_frag16); hipFree(kv.second.w_frag_halo); hipFree(kv.second.bias); hipFree(kv.second.oscale); }
    c->conv.clear();
    c->finalized = false;
    for (int i = 0; i < h.n_conv; ++i) {
        if (p + sizeof(PackEntry) > end) return fail(SMK_E_WEIGHT, "smk_import_packed: truncated entry table");
        PackEntry en;
        memcpy(&en, p, sizeof(en)); p += sizeof(en);
        en.id[sizeof(en.id) - 1] = 0;
        const int en_kw = en.kw ? en.kw : en.k;
        if (en.rows < 1 || en.Kpad < 1 || en.rows % NPAD_ALIGN || en.Kpad % KPAD_ALIGN || en.K > en.Kpad ||
            en.K < 1 || en.Ci < 8 || en.Ci % 8 || en.k < 1 || en.k > 15 || en_kw < 1 || en_kw > 15 ||
            en.K != en.k * en_kw * en.Ci || en.groups < 1 || en.group_rows < 1 || en.group_rows % NPAD_ALIGN ||
            (long)en.groups * en.group_rows != en.rows || en.N < 1 || en.N > en.group_rows || en.alg_k < 0 ||
            en.alg_k > en.K || (long)en.rows * en.Kpad > (1L << 28))
            return fail(SMK_E_WEIGHT, "smk_import_packed: entry %s has bad geometry", en.id);
        const size_t wb = (size_t)en.rows * en.Kpad * esize(c->dtype), bb = (size_t)en.rows * 4;
        if (p + wb + bb > end) return fail(SMK_E_WEIGHT, "smk_import_packed: truncated data of %s", en.id);
        PackedConv pc;
        pc.N = en.N; pc.rows = en.rows; pc.group_rows = en.group_rows; pc.groups = en.groups;
        pc.Ci = en.Ci; pc.k = en.k; pc.K = en.K; pc.Kpad = en.Kpad; pc.kw = en.kw; pc.alg_k = en.alg_k;
        HIPCHK(hipMalloc(&pc.w, wb));
        HIPCHK(hipMemcpy(pc.w, p, wb, hipMemcpyHostToDevice));
        {   // derived copies (not stored in the blob): chunk-major for the halo kernel, fragment order for conv_wreg_kernel
            std::vector<float> rf((size_t)en.rows * en.Kpad);
            if (c->dtype == DT_F16) { const _Float16 *h = (const _Float16 *)p; for (size_t i = 0; i < rf.size(); ++i) rf[i] = (float)h[i]; }
            else memcpy(rf.data(), p, wb);
            if (pc.groups == 1) CHK(upload_halo_pack(pc, rf, c->dtype));
            CHK(upload_frag_pack(pc, rf, c->dtype));
        }
        p += wb;
        HIPCHK(hipMalloc((void **)&pc.bias, bb));
        HIPCHK(hipMemcpy(pc.bias, p, bb, hipMemcpyHostToDevice)); p += bb;
        c->conv[en.id] = pc;
    }
    // every convolution the variant launches must be present
    static const char *need_all[] = {"stem", "adjust", "conv_kernel", "conv_search", "head0", "cls3", "loc3", "l1.0.ds", "l3.5.c3"};
    for (const char *id : need_all)
        if (!c->conv.count(id)) return fail(SMK_E_WEIGHT, "smk_import_packed: blob lacks %s", id);
    if (c->variant != SMK_VARIANT_RPN && !c->conv.count("mask3")) return fail(SMK_E_WEIGHT, "smk_import_packed: blob lacks mask3");
    if (c->variant == SMK_VARIANT_SHARP && (!c->conv.count("deconv") || !c->conv.count("post2")))
        return fail(SMK_E_WEIGHT, "smk_import_packed: blob lacks the Refine convolutions");
    c->finalized = true;
    c->template_B = 0;
    c->track_B = 0;
    return 0;
}

int smk_seq_status(smk_ctx *c, int *grid, int *err) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    const int rc = seq_health(c);                        // picks up a failure of the work that has just drained
    if (grid) *grid = c->seq_grid;
    if (err) *err = c->seq_fail;
    if (rc) return rc;
    if (c->seq_fail && c->seq_fail != 3)
        return fail(SMK_E_STATE, "conv_seq_kernel reported %s earlier; persistent sequences are off for this context",
                    c->seq_fail == 1 ? "an uneven distribution of workgroups over the XCDs" : "a team-barrier time-out");
    return 0;
}

int smk_seq_sync_check(smk_ctx *c, void *stream, int *synced) {
    if (synced) *synced = 0;
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    if (!c->seq_pending) return 0;                       // no sequence launch in flight: nothing to wait for, nothing to check
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (synced) *synced = 1;
    c->seq_pending = false;
    return seq_health(c);
}

int smk_set_graph_mode(smk_ctx *c, int enable) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    c->graph_mode = enable != 0;
    return 0;
}

int smk_template(smk_ctx *c, const float *z, int B, void *stream) {
    if (!c || !z) return fail(SMK_E_ARG, "smk_template: null argument");
    if (!c->finalized) return fail(SMK_E_STATE, "smk_template: weights not finalized");
    if (B < 1 || B > c->maxB) return fail(SMK_E_ARG, "smk_template: batch %d not in [1,%d]", B, c->maxB);
    HIPCHK(hipSetDevice(c->device));
    CHK(seq_health(c));
    hipStream_t s = (hipStream_t)stream;
    CHK(pipe_join(c, s, true));
    c->parity_now = c->last_parity = 0;
    GraphKey key{0, B, 0, {z}};
    int rc = run_maybe_graph(c, key, s, [&](hipStream_t st) { return seq_template(c, z, B, st); });
    if (rc) return rc;
    c->template_B = B;
    c->track_B = 0;
    return 0;
}

int smk_track(smk_ctx *c, const float *x, int B, int flags, float *cls, float *loc, float *mask, void *stream) {
    if (!c || !x || !cls || !loc) return fail(SMK_E_ARG, "smk_track: null argument");
    if (!c->finalized) return fail(SMK_E_STATE, "smk_track: weights not finalized");
    if (c->template_B == 0) return fail(SMK_E_STATE, "smk_track: smk_template has not been called");
    if (B != c->template_B)
        return fail(SMK_E_ARG, "smk_track: batch %d != template batch %d (models/rpn.py:33 requires equality)", B,
                    c->template_B);
    if ((flags & SMK_TRACK_MASK) && c->variant == SMK_VARIANT_RPN)
        return fail(SMK_E_ARG, "smk_track: the rpn variant has no mask branch");
    if ((flags & SMK_TRACK_MASK) && !(flags & SMK_TRACK_NO_MASK_HEAD) && !mask)
        return fail(SMK_E_ARG, "smk_track: mask_out is NULL");
    HIPCHK(hipSetDevice(c->device));
    CHK(seq_health(c));
    hipStream_t s = (hipStream_t)stream;
    CHK(pipe_join(c, s, true));
    c->parity_now = c->last_parity = 0;
    GraphKey key{1, B, flags, {x, cls, loc, mask}};
    int rc = run_maybe_graph(c, key, s, [&](hipStream_t st) { return seq_track(c, x, B, flags, cls, loc, mask, st); });
    if (rc) return rc;
    c->track_B = (flags & SMK_TRACK_MASK) ? B : 0;
    return 0;
}

int smk_refine(smk_ctx *c, const int32_t *pos, int on_device, int B, float *out, void *stream) {
    if (!c || !pos || !out) return fail(SMK_E_ARG, "smk_refine: null argument");
    if (c->variant != SMK_VARIANT_SHARP) return fail(SMK_E_ARG, "smk_refine: only the sharp variant has a Refine module");
    if (c->track_B == 0) return fail(SMK_E_STATE, "smk_refine: needs a preceding smk_track with SMK_TRACK_MASK");
    if (B != c->track_B) return fail(SMK_E_ARG, "smk_refine: batch %d != tracked batch %d", B, c->track_B);
    HIPCHK(hipSetDevice(c->device));
    CHK(seq_health(c));
    hipStream_t s = (hipStream_t)stream;
    CHK(pipe_join(c, s, true));
    c->parity_now = c->last_parity;          // the copy of p0 / p1 the last tracked frame left its features in
    if (!on_device) {
        for (int i = 0; i < 2 * B; ++i)
            if (pos[i] < 0 || pos[i] >= 25) return fail(SMK_E_ARG, "smk_refine: pos[%d]=%d outside [0,25)", i, pos[i]);
        HIPCHK(hipMemcpyAsync(c->pos_dev, pos, sizeof(int) * 2 * B, hipMemcpyHostToDevice, s));
    } else {
        HIPCHK(hipMemcpyAsync(c->pos_dev, pos, sizeof(int) * 2 * B, hipMemcpyDeviceToDevice, s));
    }
    GraphKey key{2, B, c->parity_now, {out}};
    return run_maybe_graph(c, key, s, [&](hipStream_t st) { return seq_refine(c, B, out, st); });
}

int smk_tune(const char *key, int value) {
    if (!key) return fail(SMK_E_ARG, "smk_tune: key is NULL");
    if (!strcmp(key, "xcd_mode")) g_tune.xcd_mode = value;
    else if (!strcmp(key, "force_tile")) { if (value < 0 || value > 5) return fail(SMK_E_ARG, "force_tile 0..5"); g_tune.force_tile = value; }
    else if (!strcmp(key, "min_blocks_x16")) g_tune.min_blocks_x16 = value;
    else if (!strcmp(key, "concurrency")) g_concurrency_default = value;
    else if (!strcmp(key, "stages")) { if (value != 0 && (value < 2 || value > 4)) return fail(SMK_E_ARG, "stages 0|2|3|4"); g_tune.stages = value; }
    else if (!strcmp(key, "merge")) { if (value < 0 || value > 2) return fail(SMK_E_ARG, "merge 0..2"); g_tune.merge = value; }
    else if (!strcmp(key, "merge_max_batch")) g_tune.merge_max_batch = value;
    else if (!strcmp(key, "seq_spoll")) g_tune.seq_spoll = value != 0;
    else if (!strcmp(key, "rf_wreg")) g_tune.rf_wreg = value;
    else if (!strcmp(key, "seq_fuse3")) {
        if (value < 0 || value > 2) return fail(SMK_E_ARG, "seq_fuse3 0..2");
#ifndef SMK_MEASURE
        if (value) return fail(SMK_E_ARG, "seq_fuse3: the triple routine (measured a wash) is only in a library built with `make MEASURE=1`");
#endif
        g_tune.seq_fuse3 = value;
    }
    else if (!strcmp(key, "nchw_tn_major")) g_tune.nchw_tn_major = value != 0;
    else if (!strcmp(key, "chain_mask")) g_tune.chain_mask = value != 0;
    else if (!strcmp(key, "wreg")) { if (value < 0 || value > 7) return fail(SMK_E_ARG, "wreg 0..7"); g_tune.wreg = value; }
    else if (!strcmp(key, "seq")) g_tune.seq = value != 0;
    else if (!strcmp(key, "ablate")) {
#ifdef SMK_MEASURE
        g_tune.ablate = value & 127;
#else
        if (value) return fail(SMK_E_ARG, "ablate: the measurement kernels are only in a library built with `make MEASURE=1`");
#endif
    }
    else if (!strcmp(key, "seq_kstag")) { if (value < 0 || value > 2) return fail(SMK_E_ARG, "seq_kstag 0|1|2"); g_tune.seq_kstag = value; }
    else if (!strcmp(key, "seq_deep")) {
#ifndef SMK_MEASURE
        if (value) return fail(SMK_E_ARG, "seq_deep: the deep-ring measurement tile is only in a library built with `make MEASURE=1`");
#endif
        g_tune.seq_deep = value != 0;
    }
    else if (!strcmp(key, "corr_head")) g_tune.corr_head = value != 0;
    else if (!strcmp(key, "pair_launch")) { if (value < 0 || value > 3) return fail(SMK_E_ARG, "pair_launch 0..3"); g_tune.pair_launch = value; }
    else if (!strcmp(key, "rf_tile2")) { if (value < 0 || value > 5) return fail(SMK_E_ARG, "rf_tile2 0..5"); g_tune.rf_tile2 = value; }
    else if (!strcmp(key, "seq_pair2d")) {
        if (value < 0 || value > 2) return fail(SMK_E_ARG, "seq_pair2d 0..2");
#ifndef SMK_MEASURE
        if (value) return fail(SMK_E_ARG, "seq_pair2d: the pair split over two CUs (measured a wash) is only in a library built with `make MEASURE=1`");
#endif
        g_tune.seq_pair2d = value;
    }
    else if (!strcmp(key, "seq_fuse")) { if (value < 0 || value > 3) return fail(SMK_E_ARG, "seq_fuse 0..3"); g_tune.seq_fuse = value; }
    else if (!strcmp(key, "seq_ds128")) g_tune.seq_ds128 = value != 0;
    else if (!strcmp(key, "seq_halo")) g_tune.seq_halo = value != 0;
    else if (!strcmp(key, "seq_kstag_mask")) g_tune.seq_kstag_mask = value & 7;
    else if (!strcmp(key, "res_nt")) g_tune.res_nt = value != 0;
    else if (!strcmp(key, "seq_tall")) { if (value < 0 || value > 2) return fail(SMK_E_ARG, "seq_tall 0|1|2"); g_tune.seq_tall = value; }
    else if (!strcmp(key, "seq_first_stage")) {
        if (value < 0 || value > 3) return fail(SMK_E_ARG, "seq_first_stage 0..3");
#ifndef SMK_MEASURE
        if (value == 0) return fail(SMK_E_ARG, "seq_first_stage 0 (layer1 inside the sequence: measured 140 us against 96) is only in a library built with `make MEASURE=1`");
#endif
        g_tune.seq_first_stage = value;
    }
    else if (!strcmp(key, "seq_min_batch")) { if (value < 1) return fail(SMK_E_ARG, "seq_min_batch >= 1"); g_tune.seq_min_batch = value; }
    else if (!strcmp(key, "seq_max_batch")) { if (value < 1) return fail(SMK_E_ARG, "seq_max_batch >= 1"); g_tune.seq_max_batch = value; }
    else if (!strcmp(key, "seq_extra_batch")) g_tune.seq_extra_batch = value;
    else if (!strcmp(key, "seq_mult_max")) { if (value < 0) return fail(SMK_E_ARG, "seq_mult_max >= 0"); g_tune.seq_mult_max = value; }
    else if (!strcmp(key, "wreg_stages")) { 
#ifdef SMK_MEASURE
        if (value != 0 && (value < 3 || value > 8)) return fail(SMK_E_ARG, "wreg_stages 0|3..8");
#else
        if (value != 0 && value != 3 && value != 4) return fail(SMK_E_ARG, "wreg_stages 0|3|4 (8 = eight k-steps ahead on every tile shape: measured slower, `make MEASURE=1` builds only)");
#endif
        g_tune.wreg_stages = value; }
    else if (!strcmp(key, "chain")) g_tune.chain = value != 0;
    else if (!strcmp(key, "halo_db")) g_tune.halo_db = value != 0;
    else if (!strcmp(key, "ksplit")) { if (value != 0 && value != 1 && value != 2 && value != 4) return fail(SMK_E_ARG, "ksplit 0|1|2|4"); g_tune.ksplit = value; }
    else if (!strcmp(key, "halo")) { if (value != 0 && value != 1 && value != 64 && value != 128) return fail(SMK_E_ARG, "halo 0|1|64|128"); g_tune.halo = value; }
    else if (!strcmp(key, "xc_full")) { if (value < 0 || value > 2) return fail(SMK_E_ARG, "xc_full 0|1|2"); g_tune.xc_full = value; }
    else if (!strcmp(key, "stem_fused")) g_tune.stem_fused = value != 0;
    else if (!strcmp(key, "l1_fused")) g_tune.l1_fused = value != 0;
    else if (!strcmp(key, "xc_ch")) { if (value != 32 && value != 64) return fail(SMK_E_ARG, "xc_ch 32|64"); g_tune.xc_ch = value; }
    else if (!strcmp(key, "buf_lds")) g_tune.buf_lds = value != 0;
    else if (!strcmp(key, "a_stage")) g_tune.a_stage = value != 0;
    else if (!strcmp(key, "wreg_policy")) { if (value != 0 && value != 1) return fail(SMK_E_ARG, "wreg_policy 0|1"); g_tune.wreg_policy = value; }
    else if (!strcmp(key, "npw")) { if (value != 2 && value != 4) return fail(SMK_E_ARG, "npw 2|4"); g_tune.npw = value; }
    else if (!strcmp(key, "mask_overlap")) g_tune.mask_overlap = value != 0;
#ifdef SMK_MEASURE
    else if (!strcmp(key, "pipe_eager")) g_tune.pipe_eager = value & 3;
    else if (!strcmp(key, "pipe_join")) g_tune.pipe_join = value != 0;
    else if (!strcmp(key, "pipe_two_form")) { if (value < 0 || value > 2) return fail(SMK_E_ARG, "pipe_two_form 0..2"); g_tune.pipe_two_form = value; }
    else if (!strcmp(key, "pipe_sig")) { if (value < 0 || value > 2) return fail(SMK_E_ARG, "pipe_sig 0..2"); g_tune.pipe_sig = value; }
#else
    else if (!strcmp(key, "pipe_eager") || !strcmp(key, "pipe_join") || !strcmp(key, "pipe_two_form") || !strcmp(key, "pipe_sig")) {
        const int dflt = !strcmp(key, "pipe_eager") ? 0 : (!strcmp(key, "pipe_join") ? 1 : (!strcmp(key, "pipe_two_form") ? 1 : 2));
        if (value != dflt) return fail(SMK_E_ARG, "%s: the measured alternatives of the pipelined step are only in a library built with `make MEASURE=1`", key);
    }
#endif
    else if (!strcmp(key, "wreg96")) g_tune.wreg96 = value != 0;
    else if (!strcmp(key, "main_prio")) { if (value < 0 || value > 3) return fail(SMK_E_ARG, "main_prio 0..3"); g_tune.main_prio = value; }
    else if (!strcmp(key, "x3_fused")) g_tune.x3_fused = value != 0;          // (read when a split-operand context packs its weights)
    else if (!strcmp(key, "pipe_prio")) {
#ifndef SMK_MEASURE
        if (value != 0) return fail(SMK_E_ARG, "pipe_prio: measured slower in both directions; only in a library built with `make MEASURE=1`");
#endif
        if (value < 0 || value > 2) return fail(SMK_E_ARG, "pipe_prio 0|1|2");
        g_tune.pipe_prio = value;
    }
    else if (!strcmp(key, "wreg32")) { if (value < 0 || value > 4096) return fail(SMK_E_ARG, "wreg32 0..4096 (64x64 tile count below which 32x64 tiles are used)"); g_tune.wreg32 = value; }
    else if (!strcmp(key, "front_occ1")) {
#ifdef SMK_MEASURE
        g_tune.front_occ1 = value & 3;
#else
        if (value) return fail(SMK_E_ARG, "front_occ1: a measured loss (profiles/r06g_front_occupancy_ab.txt), only in a library built with `make MEASURE=1`");
#endif
    }
    else if (!strcmp(key, "seq_yres")) g_tune.seq_yres = value != 0;
    else if (!strcmp(key, "seq_search")) g_tune.seq_search = value != 0;
    else if (!strcmp(key, "pp")) { if (value < 0 || value > 2) return fail(SMK_E_ARG, "pp 0..2"); g_tune.pp = value; }
    else if (!strcmp(key, "pipe_late")) g_tune.pipe_late = value != 0;
    else if (!strcmp(key, "nt_store")) g_tune.nt_store = value != 0;
    else if (!strcmp(key, "prio")) { if (value < -1 || value > 3) return fail(SMK_E_ARG, "prio -1..3"); g_tune.prio = value; }
    else if (!strcmp(key, "kt")) { if (value != 0 && value != 128 && value != 256) return fail(SMK_E_ARG, "kt 0|128|256"); g_tune.kt = value; }
    else return fail(SMK_E_ARG, "smk_tune: unknown key %s", key);
    return 0;
}

int smk_tune_get(const char *key, int *value) {
    if (!key || !value) return fail(SMK_E_ARG, "smk_tune_get: null argument");
    if (!strcmp(key, "measure_build")) {              // 1: built with `make MEASURE=1` (the K-loop ablation kernels are present)
#ifdef SMK_MEASURE
        *value = 1;
#else
        *value = 0;
#endif
        return 0;
    }
    static const struct { const char *name; int *slot; } knobs[] = {
        {"seq_fused_last", &g_seq_fused_last}, {"seq_yres_last", &g_seq_yres_last},
        {"xcd_mode", &g_tune.xcd_mode}, {"force_tile", &g_tune.force_tile}, {"min_blocks_x16", &g_tune.min_blocks_x16},
        {"concurrency", &g_concurrency_default}, {"stages", &g_tune.stages}, {"merge", &g_tune.merge}, {"merge_max_batch", &g_tune.merge_max_batch}, {"seq_spoll", &g_tune.seq_spoll}, {"rf_wreg", &g_tune.rf_wreg}, {"seq_fuse3", &g_tune.seq_fuse3}, {"seq_fused3_last", &g_seq_fused3_last},
        {"nchw_tn_major", &g_tune.nchw_tn_major}, {"chain_mask", &g_tune.chain_mask}, {"wreg", &g_tune.wreg},
        {"seq", &g_tune.seq}, {"ablate", &g_tune.ablate}, {"seq_tall", &g_tune.seq_tall}, {"seq_kstag", &g_tune.seq_kstag},
        {"seq_deep", &g_tune.seq_deep}, {"seq_fuse", &g_tune.seq_fuse}, {"seq_pair2d", &g_tune.seq_pair2d}, {"corr_head", &g_tune.corr_head}, {"pair_launch", &g_tune.pair_launch}, {"rf_tile2", &g_tune.rf_tile2}, {"seq_ds128", &g_tune.seq_ds128}, {"seq_halo", &g_tune.seq_halo}, {"seq_kstag_mask", &g_tune.seq_kstag_mask}, {"res_nt", &g_tune.res_nt},
        {"seq_first_stage", &g_tune.seq_first_stage}, {"seq_min_batch", &g_tune.seq_min_batch},
        {"seq_max_batch", &g_tune.seq_max_batch}, {"seq_extra_batch", &g_tune.seq_extra_batch}, {"seq_mult_max", &g_tune.seq_mult_max}, {"wreg_stages", &g_tune.wreg_stages}, {"chain", &g_tune.chain},
        {"halo_db", &g_tune.halo_db}, {"ksplit", &g_tune.ksplit}, {"halo", &g_tune.halo}, {"xc_ch", &g_tune.xc_ch}, {"xc_full", &g_tune.xc_full}, {"stem_fused", &g_tune.stem_fused}, {"l1_fused", &g_tune.l1_fused},
        {"buf_lds", &g_tune.buf_lds}, {"a_stage", &g_tune.a_stage}, {"npw", &g_tune.npw}, {"wreg_policy", &g_tune.wreg_policy}, {"mask_overlap", &g_tune.mask_overlap}, {"pipe_eager", &g_tune.pipe_eager}, {"pipe_join", &g_tune.pipe_join}, {"wreg96", &g_tune.wreg96}, {"main_prio", &g_tune.main_prio}, {"x3_fused", &g_tune.x3_fused}, {"pipe_prio", &g_tune.pipe_prio}, {"wreg32", &g_tune.wreg32}, {"pp", &g_tune.pp}, {"front_occ1", &g_tune.front_occ1}, {"seq_yres", &g_tune.seq_yres}, {"seq_search", &g_tune.seq_search}, {"pipe_late", &g_tune.pipe_late}, {"pipe_two_form", &g_tune.pipe_two_form}, {"pipe_sig", &g_tune.pipe_sig},
        {"nt_store", &g_tune.nt_store}, {"prio", &g_tune.prio}, {"kt", &g_tune.kt}};
    for (const auto &k : knobs)
        if (!strcmp(key, k.name)) { *value = *k.slot; return 0; }
    return fail(SMK_E_ARG, "smk_tune_get: unknown key %s", key);
}

int smk_profile(smk_ctx *c, int enable) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    c->prof_recs.clear();
    c->prof_pool_next = 0;
    if (enable && c->prof_pool.empty()) {
        c->prof_pool.resize(2 * 1024);
        for (auto &ev : c->prof_pool) HIPCHK(hipEventCreate(&ev));
        HIPCHK(hipDeviceSynchronize());
    }
    c->prof = enable != 0;
    c->prof_merge = enable == 2;
    return 0;
}

int smk_profile_dump(smk_ctx *c, char *buf, int cap) {
    if (!c || !buf || cap < 64) return fail(SMK_E_ARG, "smk_profile_dump: bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    struct Agg { std::string kernel; double ms = 0, flop = 0, bytes = 0, ext = 0; int calls = 0; };
    std::vector<std::pair<std::string, Agg>> order;
    std::map<std::string, size_t> idx;
    for (auto &r : c->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
        auto it = idx.find(r.id);
        if (it == idx.end()) { idx[r.id] = order.size(); order.push_back({r.id, Agg()}); it = idx.find(r.id); }
        Agg &a = order[it->second].second;
        a.kernel = r.kernel; a.ms += ms; a.flop += r.flop; a.bytes += r.bytes; a.ext += r.ext_bytes; a.calls++;
    }
    c->prof_recs.clear();
    c->prof_pool_next = 0;
    std::string js = "[";
    for (size_t i = 0; i < order.size(); ++i) {
        char line[512];
        const Agg &a = order[i].second;
        snprintf(line, sizeof(line), "%s{\"id\":\"%s\",\"kernel\":\"%s\",\"calls\":%d,\"ms\":%.6f,\"flop\":%.6e,\"bytes\":%.6e,\"ext_bytes\":%.6e}",
                 i ? "," : "", order[i].first.c_str(), a.kernel.c_str(), a.calls, a.ms, a.flop, a.bytes, a.ext);
        js += line;
    }
    js += "]";
    if ((int)js.size() + 1 > cap) return fail(SMK_E_ARG, "smk_profile_dump: buffer too small (%zu needed)", js.size() + 1);
    memcpy(buf, js.c_str(), js.size() + 1);
    return 0;
}

static void fill_decode_params(smk_ctx *c, const float *cls, const float *loc, int B, const double *target_wh, int *pos_out,
                               double *box_out, DecodeParams &p);

static int seq_decode(smk_ctx *c, const float *cls, const float *loc, int B, const double *target_wh, int *pos_out,
                      double *box_out, hipStream_t s) {
    DecodeParams p;
    fill_decode_params(c, cls, loc, B, target_wh, pos_out, box_out, p);
    if (c->ring_in_step && c->ring_box && box_out) {     // result ring: the box goes to the ring from the decode launch itself
        p.ring_box = c->ring_box; p.ring_cursor = c->ring_cursor; p.ring_done = (unsigned *)(c->ring_cursor + 1);
        p.ring_rows = c->ring_rows; p.ring_advance = 1;
        if (c->ring_step_refine) {
            // A Refine launch follows and commits the frame (cursor [0]).  With frame steps pipelined two deep this launch runs BEFORE the
            // previous frame's chain has done so, so the box rows follow their own cursor ([2], arrivals [3]) in every mode; both cursors
            // advance once per frame whatever mix of step forms a caller uses.
            p.ring_cursor = c->ring_cursor + 2; p.ring_done = (unsigned *)(c->ring_cursor + 3);
        } else {
            p.ring_also = c->ring_cursor + 2;          // no Refine launch: this one commits the frame and keeps the box cursor level
        }
    }
    if (c->pipe_mark_fold) { p.mark = c->pipe_cnt + 2; p.mark_arrived = c->pipe_cnt + 4; }     // pipelined step: the tail's gate waits for this launch
    ProfScope ps(c, s, "decode", "decode", 0.0, (double)B * 30 * 625 * 4);
    if (launch_decode(p, s)) return fail(SMK_E_HIP, "decode launch failed");
    return 0;
}

static void fill_decode_params(smk_ctx *c, const float *cls, const float *loc, int B, const double *target_wh, int *pos_out,
                               double *box_out, DecodeParams &p) {
    memset(&p, 0, sizeof(p));
    p.cls = cls; p.loc = loc; p.target_wh = target_wh; p.window = c->window_dev;
    p.pos_out = pos_out; p.box_out = box_out;
    {   // [maxB][8] f64 | [maxB][8][8] f64 | [maxB][8] i32 | [maxB] u32
        char *q = (char *)c->dec_scratch;
        p.part_val = (double *)q;             q += (size_t)c->maxB * 8 * 8;
        p.part_box = (double *)q;             q += (size_t)c->maxB * 64 * 8;
        p.part_idx = (int *)q;                q += (size_t)c->maxB * 8 * 4;
        p.arrived = (unsigned *)q;
    }
    p.B = B; p.A = 5; p.S = 25; p.stride = c->anchor_stride;
    for (int i = 0; i < 5; ++i) { p.anchor_w[i] = c->anchor_w[i]; p.anchor_h[i] = c->anchor_h[i]; }
    p.penalty_k = c->penalty_k; p.window_influence = c->window_influence;
}

int smk_set_decode_params(smk_ctx *c, const float *anchor_wh, int n_anchor, int stride, double penalty_k,
                          double window_influence) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    if (anchor_wh) {
        if (n_anchor != 5) return fail(SMK_E_ARG, "smk_set_decode_params: kernels are specialised for 5 anchors");
        for (int i = 0; i < 5; ++i) { c->anchor_w[i] = anchor_wh[2 * i]; c->anchor_h[i] = anchor_wh[2 * i + 1]; }
    }
    if (stride > 0) c->anchor_stride = stride;
    c->penalty_k = penalty_k;
    c->window_influence = window_influence;
    CHK(pipe_quiesce(c));
    for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second);   // hp are baked into captured launches
    c->graphs.clear();
    c->graph_used.clear();
    return 0;
}

int smk_decode(smk_ctx *c, const float *cls, const float *loc, int B, const double *target_wh, int32_t *pos_out,
               double *box_out, void *stream) {
    if (!c || !cls || !loc || !target_wh) return fail(SMK_E_ARG, "smk_decode: null argument");
    if (B < 1 || B > c->maxB) return fail(SMK_E_ARG, "smk_decode: batch %d not in [1,%d]", B, c->maxB);
    HIPCHK(hipSetDevice(c->device));
    CHK(pipe_join(c, (hipStream_t)stream, true));     // (a tail in flight reads the ctx-internal position)
    return seq_decode(c, cls, loc, B, target_wh, pos_out ? pos_out : c->pos_dev, box_out, (hipStream_t)stream);
}

// the part of a frame step that feeds the NEXT frame (its crop depends on the decoded box only, tools/test.py:240-250,302-308):
// layer2 .. decode; `phase` = PH_ALL with the front end in front of it (serial step) or PH_BACK behind a front graph
static int step_track_decode(smk_ctx *c, const float *x, int B, int flags, const double *target_wh, float *cls, float *loc,
                             float *mask, double *box_out, float *refine_out, hipStream_t st, bool defer_mask_join, int phase) {
    // sharp fp16 with Refine: the mask head rides in the Refine chain launch (see chain_mask_kernel)
    c->have_deferred_mask = false;
    c->defer_mask_req = refine_out && mask && (flags & SMK_TRACK_MASK) && !(flags & SMK_TRACK_NO_MASK_HEAD) &&
                        kdtype(c->dtype) == DT_F16 && g_tune.chain && g_tune.chain_mask && !parallel_ok(c) &&
                        (B <= 16 ||   // measured (profiles/r02_chain_mask_ab.txt): B=8 -6.6 %, B=1 -2 %, B=64 +1 % (64 chain workgroups)
                         (c->pipe_two && PIPE_TWO_FORM == 1));      // depth-2 pipelining, form 1: the tail's first part launches it (step_tail)
    int rc2 = seq_track(c, x, B, flags, cls, loc, mask, st, defer_mask_join, phase);
    c->defer_mask_req = false;
    CHK(rc2);
    // result ring (smk_set_result_ring): the decode launch writes the box row, the Refine chain launch the fp16 logits and
    // advances the cursor; only a Refine that does NOT end in the chain kernel (fp32, smk_tune chain = 0) needs the
    // stand-alone commit launch for its logits
    c->ring_in_step = c->ring_rows > 0;
    c->ring_step_refine = refine_out != nullptr && c->ring_ref != nullptr;
    c->ring_ref_folded = false;
    int rcd = seq_decode(c, cls, loc, B, target_wh, c->pos_dev, box_out, st);
    c->ring_in_step = false;
    return rcd;
}

// the part nothing on the device waits for (tools/test.py:257-284: the mask is an output): Refine at the decoded positions
// (+ the 63x63 mask head when the chain launch carries it) and the ring row's logits
static int step_tail(smk_ctx *c, int B, float *mask, double *box_out, float *refine_out, hipStream_t st, int part = 0) {
    c->ring_in_step = c->ring_rows > 0;
    c->ring_step_refine = refine_out != nullptr && c->ring_ref != nullptr;
    if (part == 1 && c->have_deferred_mask && PIPE_TWO_FORM == 1) {
        // depth-2 pipelining: the 63x63 mask head FIRST (it needs head0 only) -- as its own launch beside the next frame's front end; inside
        // the chain launch of part 2 its 640 tiles would take the CUs from that frame's conv_search (measured: 67 instead of 32 us,
        // profiles/r05j_depth2_chain_mask_beside_heads.txt); the chain alone (one workgroup per stream) runs there for free
        c->have_deferred_mask = false;
        Act h0 = act(c, "head0", 25, 25, 256 * nbranch(c));
        ConvOpt om; om.nchw_out = mask; om.cin_off = 512;
        CHK(run_conv(c, "mask3", h0, nullptr, B, om, st));
    }
    int rcd = refine_out ? seq_refine(c, B, refine_out, st, part) : 0;
    c->ring_in_step = false;
    CHK(rcd);
    if (part == 1) return 0;
    if (c->have_deferred_mask) {                 // the chain launch did not take it (timing aid on, ...): its own launch
        c->have_deferred_mask = false;
        Act h0 = act(c, "head0", 25, 25, 256 * nbranch(c));
        ConvOpt om; om.nchw_out = mask; om.cin_off = 512;
        CHK(run_conv(c, "mask3", h0, nullptr, B, om, st));
    }
    if (c->mask_join_pending) {
        c->mask_join_pending = false;
        CHK(stream_dep(c, c->side[0], st));
    }
    if (c->ring_rows > 0 && refine_out && c->ring_ref && !c->ring_ref_folded) {      // (the box row was written by the decode launch)
        RingParams rg{box_out, refine_out, nullptr, (_Float16 *)c->ring_ref, c->ring_cursor,
                      (unsigned *)(c->ring_cursor + 1), c->ring_rows, B, 127 * 127};
        ProfScope ps(c, st, "ring_commit", "ring_commit", 0.0, (double)B * (64.0 * 2 + (refine_out ? 127.0 * 127 * 6 : 0.0)));
        if (launch_ring_commit(rg, st)) return fail(SMK_E_HIP, "ring_commit launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    return 0;
}

// Pipelined frame step (smk_set_pipeline(ctx, 1)): two linear graphs per frame --
//   caller's stream:  main(f) = stem + layer1 into copy f % 2 of p0 / p1 | gate: wait for tail(f-1) | layer2 .. heads .. decode
//   side stream:      wait (event) for main(f) | tail(f) = Refine (+ mask head) at the decoded positions | completion mark
// tail(f) (small launches, low occupancy) shares the chip with the front end of frame f + 1 (bandwidth-bound); the persistent
// layer2 .. adjust launch of frame f + 1 sits behind the gate so that it still owns every CU.  The front end of f + 1 is ordered behind
// decode(f) by the caller's stream, as a tracker that crops frame f + 1 at the decoded box needs it.  Everything both sides touch is
// either written behind the gate or exists twice (p0, p1).  The join is an in-stream gate kernel (misc_kernels.hip pipe_gate_kernel),
// not an event: a cross-queue event wait on the critical path costs 15-22 us here (smk_tune "pipe_join" = 0 keeps that form --
// three graphs, front | event wait | mid -- for the A/B).
static int step_pipelined_enqueue(smk_ctx *c, const float *x, int B, int flags, const double *target_wh, float *cls, float *loc,
                                  float *mask, double *box_out, float *refine_out, hipStream_t s);
static int step_pipelined(smk_ctx *c, const float *x, int B, int flags, const double *target_wh, float *cls, float *loc,
                          float *mask, double *box_out, float *refine_out, hipStream_t s) {
    const int rc = step_pipelined_enqueue(c, x, B, flags, target_wh, cls, loc, mask, box_out, refine_out, s);
    if (rc) {
        // a main part without its tail (or the reverse) would leave the semaphores unbalanced: drain and start over
        (void)hipDeviceSynchronize();
        (void)pipe_reset_counters(c);
        c->tail_pending = false;
    }
    return rc;
}
static int step_pipelined_enqueue(smk_ctx *c, const float *x, int B, int flags, const double *target_wh, float *cls, float *loc,
                                  float *mask, double *box_out, float *refine_out, hipStream_t s) {
    const int par = c->pipe_parity;
    c->parity_now = par;
    int64_t pk, wi;
    memcpy(&pk, &c->penalty_k, 8); memcpy(&wi, &c->window_influence, 8);
    const std::vector<const void *> io{x, target_wh, cls, loc, mask, box_out, refine_out, (const void *)pk, (const void *)wi};
    const bool graphs = c->graph_mode;
    const bool gate = PIPE_JOIN != 0;
    const bool sig = gate && PIPE_SIG == 1 && c->pipe_sig;
    const bool tgate = gate && PIPE_SIG == 2;      // the tail's start is a gate kernel too (A/B)
    // where the main gate sits: in front of layer2 when that is the persistent sequence (it must own every CU), else in front of the
    // heads -- the first launches that write what the tail reads (smk_tune pipe_late = 0 keeps it in front of layer2 for the A/B)
    const bool late = gate && g_tune.pipe_late && !(seq_wanted(c, B) && !parallel_ok(c));
    // depth 2 (the persistent sequence's batches, fp16 chain path, graph replay): the tail in TWO parts.  Part 1 (window convolutions,
    // deconv, v*.2: everything that reads the kept features and the position) runs beside the next frame's front end as before; part 2
    // (the Refine chain + the mask head: one low-occupancy launch of ~50 us that only reads part 1's outputs and head0) waits for the
    // next frame's persistent launch to LEAVE and runs beside that frame's heads (conv_search / corr_head / decode leave 40-200 CUs
    // idle) -- it is launched by the NEXT smk_step, or without its gate by whatever joins the pipeline first.
    const bool two = c->pipe_depth >= 2 && graphs && gate && tgate && !late && !sig && refine_splittable(c, B) && !PIPE_EAGER;
    const int fl = flags | (par << 16) | (gate ? 1 << 17 : 0) | (sig ? 1 << 18 : 0) | (tgate ? 1 << 19 : 0) | (late ? 1 << 20 : 0) | (two ? 1 << 21 : 0) |
                   ((two && PIPE_TWO_FORM == 1) ? 1 << 22 : 0) | ((two && PIPE_TWO_FORM == 2) ? 1 << 23 : 0);
    const GraphKey kf{10, B, fl, io}, km{11, B, fl, io}, kt{12, B, fl, io}, kt2g{13, B, fl, io}, kt2n{14, B, fl, io};
    auto front = [&](hipStream_t st) { return run_backbone(c, x, B, 255, st, PH_FRONT); };
    auto mid = [&](hipStream_t st) { return step_track_decode(c, x, B, flags, target_wh, cls, loc, mask, box_out, refine_out, st, false, PH_BACK); };
    auto main_ = [&](hipStream_t st) {
        struct PrioScope { smk_ctx *c; PrioScope(smk_ctx *c_) : c(c_) { c->wave_prio_now = g_tune.main_prio; } ~PrioScope() { c->wave_prio_now = 0; } } prio_scope(c);
        CHK(front(st));
        if (!late) {
            if (launch_pipe_gate(c->pipe_cnt, c->seq_err, c->seq_err_hdev, st, 0, nullptr, c->pipe_cnt + 8)) return fail(SMK_E_HIP, "pipe_gate launch failed");
            c->cap_has_seq = true;          // (the gate reports through the sequence failure flag: checked like a sequence launch)
        }
        c->pipe_gate_late = late;           // ... else seq_track places it in front of the heads
        c->pipe_mark_fold = tgate;          // the decode launch's last writer is the main part's completion mark
        c->pipe_two = two;
        // the "chip is free for the previous frame's second tail part" semaphore: raised by the persistent launch's last leaving team
        // (forms 0 / 1) or by corr_head's first workgroup, i.e. behind conv_search (form 2)
        c->pipe_seq_exit = two && PIPE_TWO_FORM != 2; c->pipe_corr_sem = two && PIPE_TWO_FORM == 2; c->pipe_seq_exit_done = false;
        const int rcm = mid(st);
        c->pipe_corr_sem = false;
        c->pipe_mark_fold = false;
        c->pipe_gate_late = false;
        c->pipe_two = false;
        const bool exit_ok = c->pipe_seq_exit_done;
        c->pipe_seq_exit = false;
        CHK(rcm);
        if (two && !exit_ok && launch_pipe_done(c->pipe_cnt + 7, st)) return fail(SMK_E_HIP, "pipe_done launch failed");   // (no sequence launch took the mark)
        if (sig && launch_pipe_mark(c->pipe_sig, st)) return fail(SMK_E_HIP, "pipe_mark launch failed");
        return 0;
    };
    // part: 0 = the whole tail (depth 1), 1 / 2 = its two parts (depth 2); gated: with the gate at its head
    auto tail = [&](hipStream_t st, int part, bool gated) {
        if (gated && tgate && launch_pipe_gate(c->pipe_cnt + (part == 2 ? 7 : 2), c->seq_err, c->seq_err_hdev, st, 1, part == 2 ? nullptr : c->pipe_cnt + 8)) return fail(SMK_E_HIP, "pipe_gate launch failed");
        c->pipe_tail_fold = gate && part == 0;      // a whole tail that ends in chain_mask_kernel lets its last workgroup be the "done" mark
        c->pipe_done_folded = false;
        const int rct = step_tail(c, B, mask, box_out, refine_out, st, part);
        c->pipe_tail_fold = false;
        CHK(rct);
        // what the next frame's main gate waits for: the whole tail (depth 1) / part 1 (depth 2: part 2 of the PREVIOUS frame precedes it in this stream)
        if (gate && part != 2 && !c->pipe_done_folded && launch_pipe_done(c->pipe_cnt, st)) return fail(SMK_E_HIP, "pipe_done launch failed");
        return 0;
    };
    bool have = c->graphs.count(km) && c->graphs.count(kt);
    if (!gate) have = have && c->graphs.count(kf);
    if (two) have = have && c->graphs.count(kt2g) && c->graphs.count(kt2n);
    if (graphs && !have) {
        // captured together: the middle part hands the mask head over to the tail at capture time
        drop_graph(c, kf); drop_graph(c, km); drop_graph(c, kt); drop_graph(c, kt2g); drop_graph(c, kt2n);
        if (gate) CHK(capture_graph(c, km, main_));
        else { CHK(capture_graph(c, kf, front)); CHK(capture_graph(c, km, mid)); }
        const bool hm = c->have_deferred_mask;
        c->pipe_tail_has_mask = hm;
        if (two) {
            // (pipe_two_form 1: part 1 launches the mask head itself and part 2 is the bare chain; 0: the chain launch of part 2 carries it)
            CHK(capture_graph(c, kt, [&](hipStream_t st) { return tail(st, 1, true); }));
            c->have_deferred_mask = hm && PIPE_TWO_FORM != 1;
            CHK(capture_graph(c, kt2g, [&](hipStream_t st) { return tail(st, 2, true); }));
            c->have_deferred_mask = hm && PIPE_TWO_FORM != 1;
            CHK(capture_graph(c, kt2n, [&](hipStream_t st) { return tail(st, 2, false); }));
            c->have_deferred_mask = false;
        } else {
            CHK(capture_graph(c, kt, [&](hipStream_t st) { return tail(st, 0, true); }));
        }
        bool ok = c->graphs.count(km) && c->graphs.count(kt);
        if (two) ok = ok && c->graphs.count(kt2g) && c->graphs.count(kt2n);
        if (!ok) return fail(SMK_E_STATE, "internal: pipelined step graphs evicted while capturing");
    }
    if (!two && c->tail2_pending) CHK(pipe_flush(c));       // (the mode changed under a pending second part: launch it now)
    if (gate) {
        CHK(graphs ? launch_graph(c, km, s) : main_(s));
        if (!graphs) c->seq_pending = true;
        // (no event wait on `s`: the gate inside main(f) is the join; tail_ev stays for the serial entry points and smk_pipeline_join)
        c->tail_pending = false;
    } else {
        CHK((graphs && !(PIPE_EAGER & 1)) ? launch_graph(c, kf, s) : front(s));
        CHK(pipe_join(c, s, true));
        CHK(graphs ? launch_graph(c, km, s) : mid(s));
    }
    hipEvent_t e_dec = c->pipe_ev[c->pipe_ev_next++ % c->pipe_ev.size()];
    hipEvent_t e_tail = c->pipe_ev[c->pipe_ev_next++ % c->pipe_ev.size()];
    if (sig) {
        // the side stream's command processor polls the counter the main graph's last kernel advances: nothing is enqueued on `s`
        if (c->pipe_sig_n >= 0x7fff0000u) {                   // (every 2^31 frames: start the count over)
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemset(c->pipe_sig, 0, 8));
            c->pipe_sig_n = 0;
            return fail(SMK_E_STATE, "internal: pipelined frame counter wrapped; re-submit the frame");
        }
        HIPCHK(hipStreamWaitValue32(c->pipe_stream, c->pipe_sig, ++c->pipe_sig_n, hipStreamWaitValueGte, 0xFFFFFFFFu));
    } else if (!tgate) {
        HIPCHK(hipEventRecord(e_dec, s));
        HIPCHK(hipStreamWaitEvent(c->pipe_stream, e_dec, 0));
    }
    if (two) {
        // the side stream, in the order things happen on the device: [sequence of THIS frame has left] part 2 of the previous frame |
        // [decode of this frame] part 1 of this frame.  Every step takes exactly one count of the "sequence has left" semaphore: with
        // no part 2 pending (first step, or behind a flush) a bare gate does.
        if (c->tail2_pending) CHK(launch_graph(c, c->tail2_gated_key, c->pipe_stream));
        else if (launch_pipe_gate(c->pipe_cnt + 7, c->seq_err, c->seq_err_hdev, c->pipe_stream, 1)) return fail(SMK_E_HIP, "pipe_gate launch failed");
        CHK(launch_graph(c, kt, c->pipe_stream));
        c->tail2_pending = true;
        c->tail2_key = kt2n;
        c->tail2_gated_key = kt2g;
    } else if (graphs && !(PIPE_EAGER & 2)) CHK(launch_graph(c, kt, c->pipe_stream));
    else {
        // (eager: the capture-time hand-over of the mask head is replayed from the context, see step_track_decode)
        if (graphs) c->have_deferred_mask = c->pipe_tail_has_mask;
        CHK(tail(c->pipe_stream, 0, true));
    }
    HIPCHK(hipEventRecord(e_tail, c->pipe_stream));
    c->tail_ev = e_tail;
    c->tail_pending = true;
    c->last_parity = par;
    c->pipe_parity = par ^ 1;
    return 0;
}

int smk_step(smk_ctx *c, const float *x, int B, int flags, const double *target_wh, float *cls, float *loc,
             float *mask, double *box_out, float *refine_out, void *stream) {
    if (!c || !x || !cls || !loc || !target_wh || !box_out) return fail(SMK_E_ARG, "smk_step: null argument");
    if (!c->finalized) return fail(SMK_E_STATE, "smk_step: weights not finalized");
    if (c->template_B == 0) return fail(SMK_E_STATE, "smk_step: smk_template has not been called");
    if (B != c->template_B) return fail(SMK_E_ARG, "smk_step: batch %d != template batch %d", B, c->template_B);
    if (refine_out && (c->variant != SMK_VARIANT_SHARP || !(flags & SMK_TRACK_MASK)))
        return fail(SMK_E_ARG, "smk_step: refine needs the sharp variant and SMK_TRACK_MASK");
    if ((flags & SMK_TRACK_MASK) && c->variant == SMK_VARIANT_RPN) return fail(SMK_E_ARG, "smk_step: rpn has no mask branch");
    if ((flags & SMK_TRACK_MASK) && !(flags & SMK_TRACK_NO_MASK_HEAD) && !mask) return fail(SMK_E_ARG, "smk_step: mask_out is NULL");
    if (c->ring_rows > 0 && B != c->ring_batch)
        return fail(SMK_E_ARG, "smk_step: batch %d, but the result ring was set for batch %d (smk_set_result_ring)", B, c->ring_batch);
    HIPCHK(hipSetDevice(c->device));
    CHK(seq_health(c));
    hipStream_t s = (hipStream_t)stream;
    // pipelined: only a step with a Refine tail has something to overlap; the profiler times launches one by one; the fork / join
    // concurrency knob and a persistent sequence that includes layer1 (measurement knobs) keep the serial step; so does split-K
    // (ONE scratch + arrival-counter set per context: layer1 of frame f + 1 and the Refine convolutions of frame f would share it)
    if (c->pipe_depth > 0 && refine_out && !c->prof && !parallel_ok(c) && g_tune.seq_first_stage >= 1 && !g_tune.mask_overlap &&
        !g_tune.ksplit) {
        int rc = step_pipelined(c, x, B, flags, target_wh, cls, loc, mask, box_out, refine_out, s);
        if (rc) return rc;
        c->track_B = B;
        return 0;
    }
    CHK(pipe_join(c, s, true));
    c->parity_now = c->last_parity = 0;
    int64_t pk, wi;
    memcpy(&pk, &c->penalty_k, 8); memcpy(&wi, &c->window_influence, 8);
    GraphKey key{3, B, flags, {x, target_wh, cls, loc, mask, box_out, refine_out, (const void *)pk, (const void *)wi}};
    int rc = run_maybe_graph(c, key, s, [&](hipStream_t st) {
        CHK(step_track_decode(c, x, B, flags, target_wh, cls, loc, mask, box_out, refine_out, st, true, PH_ALL));
        return step_tail(c, B, mask, box_out, refine_out, st);
    });
    if (rc) return rc;
    c->track_B = (flags & SMK_TRACK_MASK) ? B : 0;
    return 0;
}

int smk_set_pipeline(smk_ctx *c, int depth) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    if (depth < 0 || depth > 2)
        return fail(SMK_E_ARG, "smk_set_pipeline: depth %d (0 = off, 1 = the tail beside the next frame's front end, 2 = its second part beside the next frame's heads)", depth);
    HIPCHK(hipSetDevice(c->device));
    CHK(pipe_quiesce(c));
    HIPCHK(hipDeviceSynchronize());
    c->tail_pending = false;
    if (depth > 0) {
        // The tail's gate polls on the side stream WHILE the step's own kernels run on the caller's stream.  With ONE hardware queue
        // (GPU_MAX_HW_QUEUES=1) both streams share it: the gate sits in front of the very work it waits for, stalls for its 5 s limit
        // and raises SMK_E_SEQ.  Detect, don't stall: serial steps (same bits), said once.
        const char *hq = getenv("GPU_MAX_HW_QUEUES");
        if (hq && *hq && atoi(hq) < 2) {
            static bool said = false;
            if (!said) fprintf(stderr, "siammask_hip: GPU_MAX_HW_QUEUES=%s -- the pipelined frame step needs two hardware queues; running serial steps\n", hq);
            said = true;
            depth = 0;
        }
    }
    if (depth > 0) {
        if (!c->buf.count("p0#1")) {
            CHK(alloc_buf(c, "p0#1", c->buf_elems.at("p0")));
            CHK(alloc_buf(c, "p1#1", c->buf_elems.at("p1")));
            CHK(alloc_buf(c, "p2#1", c->buf_elems.at("p2")));
        }
        if (depth >= 2 && !c->buf.count("head0#1")) CHK(alloc_buf(c, "head0#1", c->buf_elems.at("head0")));
        // (the box rows' own cursor of depth 2 starts where the shared one stands)
        if (c->ring_cursor) HIPCHK(hipMemcpy(c->ring_cursor + 2, c->ring_cursor, sizeof(int), hipMemcpyDeviceToDevice));
        if (!c->pipe_stream) {
            // A queue priority for the side stream (lowest: the tail's workgroups dispatched behind the next frame's front end) was measured in round 6
            // (profiles/r06z_side_stream_priority.txt): ANY priority other than the default -- lowest or highest -- costs +50 % per B = 8 step and x4.4 at
            // B = 1; queues of unequal priority are not arbitrated workgroup by workgroup.  The arm lives in `make MEASURE=1` builds (smk_tune pipe_prio).
#ifdef SMK_MEASURE
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // (numerically: lo >= hi, lower value = higher priority)
            if (g_tune.pipe_prio) HIPCHK(hipStreamCreateWithPriority(&c->pipe_stream, hipStreamNonBlocking, g_tune.pipe_prio == 1 ? lo : hi));
            else
#endif
            HIPCHK(hipStreamCreateWithFlags(&c->pipe_stream, hipStreamNonBlocking));
        }
        if (!c->pipe_cnt) HIPCHK(hipMalloc((void **)&c->pipe_cnt, 64));
        if (!c->pipe_sig) {
            int can = 0;
            (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
            if (can && hipExtMallocWithFlags((void **)&c->pipe_sig, 8, hipMallocSignalMemory) != hipSuccess) { c->pipe_sig = nullptr; (void)hipGetLastError(); }
        }
        CHK(pipe_reset_counters(c));
        if (c->pipe_ev.empty()) {
            c->pipe_ev.resize(16);
            for (auto &e : c->pipe_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    c->pipe_depth = depth;
    c->pipe_parity = 0;
    return 0;
}

int smk_pipeline_join(smk_ctx *c, void *stream) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    return pipe_join(c, (hipStream_t)stream, false);
}

int smk_pipeline_observe(smk_ctx *c, void *stream) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (!c->tail_pending) return 0;
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->tail_ev, 0));
    return 0;
}

int smk_set_result_ring(smk_ctx *c, double *box_ring, void *refine_ring_f16, int rows, int batch) {
    if (!c) return fail(SMK_E_ARG, "ctx is NULL");
    if (rows < 0 || (rows > 0 && !box_ring)) return fail(SMK_E_ARG, "smk_set_result_ring: rows %d / box ring %p", rows, (void *)box_ring);
    if (rows > 0 && (batch < 1 || batch > c->maxB)) return fail(SMK_E_ARG, "smk_set_result_ring: batch %d not in [1,%d]", batch, c->maxB);
    HIPCHK(hipSetDevice(c->device));
    CHK(pipe_quiesce(c));
    HIPCHK(hipDeviceSynchronize());
    c->tail_pending = false;
    // the captured step graphs carry the ring pointers (or no commit launch at all): start over
    for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second);
    c->graphs.clear();
    c->graph_used.clear();
    c->graph_has_seq.clear();
    if (!c->ring_cursor) HIPCHK(hipMalloc((void **)&c->ring_cursor, 4 * sizeof(int)));
    HIPCHK(hipMemset(c->ring_cursor, 0, 4 * sizeof(int)));
    c->ring_box = rows ? box_ring : nullptr;
    c->ring_ref = rows ? refine_ring_f16 : nullptr;
    c->ring_rows = rows;
    c->ring_batch = rows ? batch : 0;
    return 0;
}

int smk_result_ring_cursor(smk_ctx *c, int *frames_out, int reset, void *stream) {
    if (!c || !c->ring_cursor) return fail(SMK_E_STATE, "smk_result_ring_cursor: no result ring set");
    HIPCHK(hipSetDevice(c->device));
    CHK(pipe_join(c, (hipStream_t)stream, true));        // the cursor is advanced by the tail of the last pipelined step
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (frames_out) HIPCHK(hipMemcpy(frames_out, c->ring_cursor, sizeof(int), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(c->ring_cursor, 0, 4 * sizeof(int)));
    return 0;
}

int smk_debug_seq_inject(smk_ctx *c, int code) {
    if (!c || !c->seq_err || !c->seq_err_host) return fail(SMK_E_ARG, "smk_debug_seq_inject: no context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(c->seq_err, &code, sizeof(int), hipMemcpyHostToDevice));
    *(volatile int *)c->seq_err_host = code;
    return 0;
}

int smk_debug_read(smk_ctx *c, const char *name, float *dst, int *C, int *H, int *W, void *stream) {
    if (!c || !name) return fail(SMK_E_ARG, "smk_debug_read: null argument");
    const int nbt = nbranch(c);
    const int S = c->last_S ? c->last_S : 255;
    const int s0 = (S - 7) / 2 + 1, s1 = (s0 + 2 - 3) / 2 + 1, s2 = (s1 - 3) / 2 + 1;
    struct E { const char *n, *b; int h, w, cs, cn; };
    const E tab[] = {
        {"p0", "p0", s0, s0, 64, 64}, {"p1", "p1", s1, s1, 256, 256}, {"p2", "p2", s2, s2, 512, 512},
        {"p3", c->p3_buf, s2, s2, 1024, 1024}, {"search", "search", 31, 31, 256, 256}, {"zf", "zf", 7, 7, 256, 256},
        {"zk", "zk", 5, 5, 256 * nbt, 256 * nbt}, {"xs", "xs", 29, 29, 256 * nbt, 256 * nbt},
        {"corr", "corr", 25, 25, 256 * nbt, 256 * nbt}, {"head0", "head0", 25, 25, 256 * nbt, 256 * nbt},
    };
    for (auto &e : tab)
        if (!strcmp(e.n, name)) {
            if (C) *C = e.cn;
            if (H) *H = e.h;
            if (W) *W = e.w;
            if (!dst) return 0;
            if (c->last_B < 1) return fail(SMK_E_STATE, "smk_debug_read: nothing has run yet");
            CHK(pipe_join(c, (hipStream_t)stream, true));
            c->parity_now = c->last_parity;            // p0 / p1: the copy the last tracked frame wrote
            if (c->dtype == DT_F16X3) {
                // split tensors: value = hi + lo.  Whole-tensor planes, except corr / head0 (per-branch planes: 256 g + cc -> 512 g + cc)
                const bool per_branch = !strcmp(e.n, "corr") || !strcmp(e.n, "head0");
                const int ng = per_branch ? e.cn / 256 : 1, cg = e.cn / ng;
                for (int g = 0; g < ng; ++g) {
                    CvtOutParams p{act(c, e.b, e.h, e.w, X3_PLANES * e.cs).p, dst, c->last_B, cg, e.h, e.w, X3_PLANES * e.cs, g * X3_PLANES * 256, per_branch ? 256 : e.cs};
                    if (ng > 1) return fail(SMK_E_ARG, "smk_debug_read: %s of a split-operand context is read per branch: not implemented", e.n);
                    if (launch_cvt_out_x3(p, stream)) return fail(SMK_E_HIP, "cvt_out launch failed");
                }
                return 0;
            }
            CvtOutParams p{act(c, e.b, e.h, e.w, e.cs).p, dst, c->last_B, e.cn, e.h, e.w, e.cs, 0};
            if (launch_cvt_out(p, c->dtype, stream)) return fail(SMK_E_HIP, "cvt_out launch failed");
            return 0;
        }
    return fail(SMK_E_ARG, "smk_debug_read: unknown tensor %s", name);
}

// ---- per-op entry points -------------------------------------------------------------------
struct TmpBufs {
    std::vector<void *> v;
    ~TmpBufs() { for (void *p : v) hipFree(p); }
    int alloc(void **p, size_t bytes) {
        HIPCHK(hipMalloc(p, bytes + 256));
        HIPCHK(hipMemset(*p, 0, bytes + 256));
        v.push_back(*p);
        return 0;
    }
};

static int fill_geom(const smk_conv_geom *g, PackedConv &pc, Act &in, ConvOpt &o, int &Ho, int &Wo) {
    const int cin_len = g->cin_len > 0 ? g->cin_len : g->Cin;
    pc.Ci = rup(cin_len, 8);
    pc.k = g->k;
    pc.K = g->k * g->k * pc.Ci;
    pc.Kpad = rup(pc.K, KPAD_ALIGN);
    pc.groups = 1;
    pc.N = g->Cout;
    pc.group_rows = pc.rows = rup(g->Cout, NPAD_ALIGN);
    in.H = g->H; in.W = g->W; in.C = rup(g->Cin, 8);
    o.stride = g->stride; o.pad = g->pad; o.dil = g->dil; o.relu = g->relu;
    o.cin_off = g->cin_off;
    o.win = g->win != 0; o.ups = g->ups != 0;
    o.Hl = g->Hl; o.Wl = g->Wl; o.org_y = g->org_y; o.org_x = g->org_x;
    o.pos_mul = g->pos_mul; o.pos_add = g->pos_add;
    const int Hl = (o.win || o.ups) ? g->Hl : g->H, Wl = (o.win || o.ups) ? g->Wl : g->W;
    Ho = (Hl + 2 * g->pad - g->dil * (g->k - 1) - 1) / g->stride + 1;
    Wo = (Wl + 2 * g->pad - g->dil * (g->k - 1) - 1) / g->stride + 1;
    if (Ho < 1 || Wo < 1) return fail(SMK_E_ARG, "conv geometry gives empty output");
    if (g->cin_off % 8 != 0) return fail(SMK_E_ARG, "cin_off must be a multiple of 8");
    if (g->cin_off + cin_len > g->Cin) return fail(SMK_E_ARG, "channel slice out of range");
    return 0;
}

static void pack_host(const smk_conv_geom *g, const PackedConv &pc, const float *w, const float *b,
                      std::vector<float> &rows, std::vector<float> &bias) {
    const int cin_len = g->cin_len > 0 ? g->cin_len : g->Cin;
    rows.assign((size_t)pc.rows * pc.Kpad, 0.f);
    bias.assign(pc.rows, 0.f);
    std::vector<double> one(g->Cout, 1.0);
    pack_rows(rows, 0, pc.Kpad, w, one.data(), g->Cout, cin_len, g->k, pc.Ci);
    if (b) for (int n = 0; n < g->Cout; ++n) bias[n] = b[n];
}

// split-K scratch of the context-free entry points (per-op tests, smk_bench_conv): one per device, lazily
static int op_ks_scratch(smk_ctx &fake) {
    static float *part[64] = {nullptr};
    static unsigned *cnt[64] = {nullptr};
    const int d = fake.device;
    if (d < 0 || d >= 64) return 0;
    if (!part[d]) {
        HIPCHK(hipMalloc((void **)&part[d], KS_PART_FLOATS * sizeof(float)));
        HIPCHK(hipMalloc((void **)&cnt[d], KS_CNT * sizeof(unsigned)));
        HIPCHK(hipMemset(cnt[d], 0, KS_CNT * sizeof(unsigned)));
    }
    fake.ks_part = part[d];
    fake.ks_cnt = cnt[d];
    return 0;
}


int smk_op_conv2d_ex(int dtype, int algo, const smk_conv_geom *g, const float *x_dev, const float *w_host,
                     const float *b_host, const float *res_dev, const int32_t *pos_host, float *y_dev,
                     void *stream) {
    if (!g || !x_dev || !w_host || !y_dev) return fail(SMK_E_ARG, "smk_op_conv2d_ex: null argument");
    if (dtype != DT_F32 && dtype != DT_F16 && dtype != DT_F16X3) return fail(SMK_E_ARG, "bad dtype");
    // DT_F16X3 (unit parity of the split-operand convolution): x, res -> [hi | lo] planes, the pack tripled per tap ([w_hi | w_lo | w_hi]), conv_igemm_kernel's
    // (or wreg_tile's) splitting epilogue, the output read back as hi + lo.  NHWC epilogue only, whole channel range.
    const bool x3 = dtype == DT_F16X3;
    if (x3 && (((algo & 0xff) != 0 && (algo & 0xff) != 5) || g->cin_off || g->cin_len))
        return fail(SMK_E_ARG, "smk_op_conv2d_ex: split-operand convolutions: algo 0 (conv_igemm_kernel) or 5 (conv_wreg_kernel), no channel slice");
    const int ctx_dtype = dtype;
    dtype = kdtype(dtype);
    hipStream_t s = (hipStream_t)stream;
    PackedConv pc; Act in; ConvOpt o; int Ho, Wo;
    CHK(fill_geom(g, pc, in, o, Ho, Wo));
    std::vector<float> rows, bias;
    pack_host(g, pc, w_host, b_host, rows, bias);
    if (x3) {
        const int Ci0 = pc.Ci, K1 = pc.Kpad;
        pc.x3 = true;
        pc.alg_k = g->k * g->k * Ci0;
        pc.Ci = 3 * Ci0;
        pc.K = g->k * g->k * pc.Ci;
        pc.Kpad = rup(pc.K, KPAD_ALIGN);
        std::vector<float> osc;
        split_rows_x3(rows, pc.rows, K1, pc.Kpad, g->k * g->k, Ci0, osc);
        CHK(upload_oscale(pc, osc));
        in.C *= X3_PLANES;
    }
    std::vector<float> fused;
    if (x3 && g_tune.x3_fused && fuse_rows_x3(rows, pc, g->k * g->k, pc.Ci / 3, fused)) CHK(upload_packed(pc, rows, bias, dtype, &fused));
    else
    CHK(upload_packed(pc, rows, bias, dtype));
    TmpBufs tmp;
    tmp.v.push_back(pc.w); tmp.v.push_back(pc.bias);
    if (pc.w_frag) tmp.v.push_back(pc.w_frag);
    if (pc.w_frag16) tmp.v.push_back(pc.w_frag16);
    if (pc.oscale) tmp.v.push_back(pc.oscale);
    const size_t es = esize(dtype);
    CHK(tmp.alloc(&in.p, (size_t)g->B * g->H * g->W * in.C * es));
    CvtInParams ci{x_dev, in.p, g->B, g->Cin, g->H, g->W, x3 ? in.C / X3_PLANES : in.C, 0, x3 ? 1 : 0};
    if (x3 ? launch_cvt_in_x3(ci, s) : launch_cvt_in(ci, dtype, s)) return fail(SMK_E_HIP, "cvt_in launch failed");
    int *pos_dev = nullptr;
    if (pos_host) {
        CHK(tmp.alloc((void **)&pos_dev, sizeof(int) * 2 * g->B));
        HIPCHK(hipMemcpyAsync(pos_dev, pos_host, sizeof(int) * 2 * g->B, hipMemcpyHostToDevice, s));
        o.pos = pos_dev;
    }
    const int mode = algo & 0xff;
    o.tile_code = (algo >> 8) & 0xff;
    o.algo_naive = (mode == 1 || mode == 3);
    const bool nchw = (mode == 2 || mode == 3);
    if (mode == 4) {                                   // halo kernel (3x3 stride 1), BM from the tile code
        o.halo = (o.tile_code & 15) == 1 ? 128 : 64;
        CHK(upload_halo_pack(pc, rows, dtype));
        if (pc.w_frag_halo) tmp.v.push_back(pc.w_frag_halo);     // (f16: upload_halo_pack also makes the fragment-order copy)
        if (!pc.w_halo) return fail(SMK_E_ARG, "smk_op_conv2d_ex: geometry is not eligible for the halo kernel");
        tmp.v.push_back(pc.w_halo);
    }
    Act out, res;
    out.H = Ho; out.W = Wo; out.C = rup(g->Cout, 8) * (x3 ? X3_PLANES : 1);
    if (res_dev && g->res_mode) {
        if (nchw) return fail(SMK_E_ARG, "residual is not supported with the NCHW epilogue");
        res = out;
        CHK(tmp.alloc(&res.p, (size_t)g->B * Ho * Wo * out.C * es));
        CvtInParams cr{res_dev, res.p, g->B, g->Cout, Ho, Wo, x3 ? out.C / X3_PLANES : out.C, 0, x3 ? 1 : 0};
        if (x3 ? launch_cvt_in_x3(cr, s) : launch_cvt_in(cr, dtype, s)) return fail(SMK_E_HIP, "cvt_in launch failed");
        o.res = &res; o.res_mode = g->res_mode;
    }
    smk_ctx fake;
    fake.dtype = ctx_dtype;
    HIPCHK(hipGetDevice(&fake.device));
    CHK(op_ks_scratch(fake));
    ConvParams p;
    if (nchw) {
        o.nchw_out = y_dev;
        CHK(conv_params(&fake, pc, in, nullptr, g->B, o, p));
    } else {
        CHK(tmp.alloc(&out.p, (size_t)g->B * Ho * Wo * out.C * es));
        CHK(conv_params(&fake, pc, in, &out, g->B, o, p));
    }
    int rc;
    if (mode == 5) {                                   // conv_wreg_kernel, tile code 1..6 in the low tile bits
        const int wr = o.tile_code & 15;
        if (wr < 1 || wr > 8) return fail(SMK_E_ARG, "smk_op_conv2d_ex: wreg tile code 1..8");
        ConvBatch cb;
        cb.n = 1;
        cb.p[0] = p;
        rc = launch_conv_wreg_batch(cb, WREG_TILE[wr][0], WREG_TILE[wr][1], wreg_stages_from_code(o.tile_code), s);
        if (rc == 1) return fail(SMK_E_ARG, "smk_op_conv2d_ex: geometry / dtype is not eligible for conv_wreg_kernel");
    } else if (mode == 6) {                            // conv_pp_kernel (256 x 256 tiles)
        rc = dtype == DT_F16 ? launch_conv_pp(p, s) : 1;
        if (rc == 1) return fail(SMK_E_ARG, "smk_op_conv2d_ex: geometry / dtype is not eligible for conv_pp_kernel");
    } else if (o.halo) {
        ConvParams ph = p;
        ph.wgt = pc.w_halo;
        rc = launch_conv_halo(ph, dtype, o.halo, s);
        if (rc == 1) return fail(SMK_E_ARG, "smk_op_conv2d_ex: geometry is not eligible for the halo kernel");
    } else {
        rc = o.algo_naive ? launch_conv_naive(p, dtype, s)
                          : launch_conv_mfma(p, dtype, tile_from_code(o.tile_code, p, dtype), s);
    }
    if (rc) return fail(SMK_E_HIP, "conv launch failed: %s", hipGetErrorString(hipGetLastError()));
    if (!nchw) {
        CvtOutParams co{out.p, y_dev, g->B, g->Cout, Ho, Wo, out.C, 0, x3 ? out.C / X3_PLANES : 0};
        if (x3 ? launch_cvt_out_x3(co, s) : launch_cvt_out(co, dtype, s)) return fail(SMK_E_HIP, "cvt_out launch failed");
    }
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// A sequence of convolutions through conv_seq_kernel on caller-described layers (unit parity of the persistent kernel:
// every tile configuration, residual and independent-member cases; micro-benchmarks of its K loop).  fp16 only.
int smk_op_conv_seq(const smk_seq_op *ops, int n, const float *x_dev, int iters, float *usec_out, float *clk_us_out,
                    int *n_fused_out, void *stream) {
    if (!ops || !x_dev || n < 1 || n > SEQ_MAX || iters < 1) return fail(SMK_E_ARG, "smk_op_conv_seq: bad argument (1..%d layers)", SEQ_MAX);
    hipStream_t s = (hipStream_t)stream;
    const int dtype = DT_F16;
    const size_t es = esize(dtype);
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    const int grid = seq_grid_for(prop.multiProcessorCount);
    if (!grid) return fail(SMK_E_STATE, "smk_op_conv_seq: the XCD placement / occupancy check failed on this device");
    TmpBufs tmp;
    const int B = ops[0].g.B;
    Act xin;
    xin.H = ops[0].g.H; xin.W = ops[0].g.W; xin.C = rup(ops[0].g.Cin, 8);
    CHK(tmp.alloc(&xin.p, (size_t)B * xin.H * xin.W * xin.C * es));
    CvtInParams ci{x_dev, xin.p, B, ops[0].g.Cin, xin.H, xin.W, xin.C, 0};
    if (launch_cvt_in(ci, dtype, s)) return fail(SMK_E_HIP, "cvt_in launch failed");
    smk_ctx fake;
    fake.dtype = dtype;
    fake.device = dev;
    std::vector<Act> outs(n);
    std::vector<int> couts(n);
    std::vector<PackedConv> packs(n);
    SeqArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.B = B;
    a.flags = g_tune.seq_spoll ? 1 : 0;
    std::vector<std::string> ids;
    std::vector<char> locked(n, 0);
    for (int i = 0; i < n; ++i) {
        const smk_seq_op &op = ops[i];
        locked[i] = op.cfg >= 0;
        if (!op.w_host) return fail(SMK_E_ARG, "smk_op_conv_seq: layer %d has no weights", i);
        if (op.src >= i || op.res_src >= i) return fail(SMK_E_ARG, "smk_op_conv_seq: layer %d reads a later layer", i);
        if (op.g.B != B || op.g.win || op.g.ups) return fail(SMK_E_ARG, "smk_op_conv_seq: layer %d: one batch, no windows", i);
        const Act &in = op.src < 0 ? xin : outs[op.src];
        const int cin_have = op.src < 0 ? ops[0].g.Cin : couts[op.src];
        if (op.g.H != in.H || op.g.W != in.W || op.g.Cin != cin_have)
            return fail(SMK_E_ARG, "smk_op_conv_seq: layer %d geometry does not match its source", i);
        PackedConv pc; Act gin; ConvOpt o; int Ho, Wo;
        CHK(fill_geom(&op.g, pc, gin, o, Ho, Wo));
        int same = -1;                                    // a layer that names the SAME host weights re-uses the device copy
        for (int j = 0; j < i && same < 0; ++j)           // (micro-benchmarks: L2-warm weights)
            if (ops[j].w_host == op.w_host && ops[j].b_host == op.b_host && ops[j].g.Cout == op.g.Cout &&
                ops[j].g.Cin == op.g.Cin && ops[j].g.k == op.g.k && ops[j].g.cin_len == op.g.cin_len) same = j;
        if (same >= 0) {
            pc.w = packs[same].w; pc.bias = packs[same].bias; pc.w_frag = packs[same].w_frag; pc.w_frag_halo = packs[same].w_frag_halo;
        } else {
            std::vector<float> rows, bias;
            pack_host(&op.g, pc, op.w_host, op.b_host, rows, bias);
            CHK(upload_packed(pc, rows, bias, dtype));
            if (pc.k == 3) CHK(upload_halo_pack(pc, rows, dtype));       // (+ its fragment-order form: the patch-sharing tile)
            tmp.v.push_back(pc.w); tmp.v.push_back(pc.bias);
            if (pc.w_frag) tmp.v.push_back(pc.w_frag);
            if (pc.w_frag16) tmp.v.push_back(pc.w_frag16);
            if (pc.w_halo) tmp.v.push_back(pc.w_halo);
            if (pc.w_frag_halo) tmp.v.push_back(pc.w_frag_halo);
        }
        packs[i] = pc;
        outs[i].H = Ho; outs[i].W = Wo; outs[i].C = rup(op.g.Cout, 8);
        couts[i] = op.g.Cout;
        CHK(tmp.alloc(&outs[i].p, (size_t)B * Ho * Wo * outs[i].C * es));
        Act res;
        if (op.g.res_mode) {
            if (op.res_src < -1) return fail(SMK_E_ARG, "smk_op_conv_seq: layer %d wants a residual without a source", i);
            res = op.res_src < 0 ? xin : outs[op.res_src];
            if (res.H != Ho || res.W != Wo || res.C != outs[i].C)
                return fail(SMK_E_ARG, "smk_op_conv_seq: layer %d: residual shape differs from the output", i);
            o.res = &res; o.res_mode = op.g.res_mode;
        }
        ConvParams p;
        CHK(conv_params(&fake, pc, in, &outs[i], B, o, p));
        SeqLayer L;
        const int force_halo = op.cfg == SEQ_CFG_HALO128 ? 128 : (op.cfg == SEQ_CFG_HALO64 ? 64 : (op.cfg >= 0 ? -1 : 0));
        if (!seq_layer_from(p, dtype, L, force_halo)) return fail(SMK_E_ARG, "smk_op_conv_seq: layer %d cannot run inside a sequence%s", i, force_halo > 0 ? " on the patch-sharing tile" : "");
        if (op.cfg >= 0 && force_halo <= 0) {
#ifdef SMK_MEASURE
            if (op.cfg > 18) return fail(SMK_E_ARG, "smk_op_conv_seq: cfg 0..18");
#else
            if (op.cfg > 9 || (op.cfg >= 6 && op.cfg <= 8))
                return fail(SMK_E_ARG, "smk_op_conv_seq: cfg 0..5, 9 (6..8, 10..18 are measurement tiles: `make MEASURE=1`)");
#endif
            L.cfg = (signed char)op.cfg;
        }
        if (op.kstag >= 0) L.kstag = (signed char)(op.kstag != 0);
        L.sync = (signed char)(op.sync != 0);
        a.L[i] = L;
        char nm[16];
        snprintf(nm, sizeof(nm), "op%d", i);
        ids.push_back(nm);
    }
    unsigned *bar = nullptr;
    int *err = nullptr;
    unsigned long long *clk = nullptr, *clk2 = nullptr;
    CHK(tmp.alloc((void **)&bar, 8 * 32 * sizeof(unsigned)));
    CHK(tmp.alloc((void **)&err, sizeof(int)));
    CHK(tmp.alloc((void **)&clk, sizeof(unsigned long long) * (2 * SEQ_MAX + 1)));
    const char *ck = getenv("SMK_SEQ_CLK");
    const bool want2 = ck && !strcmp(ck, "2");
    if (want2) CHK(tmp.alloc((void **)&clk2, sizeof(unsigned long long) * (12 + 32) * SEQ_MAX));
    float *xch = nullptr;
    CHK(tmp.alloc((void **)&xch, SEQ_XCH_BYTES));
    a.bar = bar; a.xch = xch; a.err = err; a.err_host = nullptr; a.clk = clk; a.clk2 = clk2;
    seq_fuse_pairs(a.L, a.n, B, &locked, (grid >> 3) % 2 == 0 && (grid >> 4) <= SEQ_XCH_PAIRS);                   // what the engine does with its own lists (smk_tune "seq_fuse")
    {
        std::vector<const void *> wstd(n);
        for (int i = 0; i < n; ++i) wstd[i] = packs[i].w_frag;
        seq_fuse_triples(a.L, a.n, B, wstd.data(), &locked);
        std::vector<char> keep(n, 0);
        for (int i = 0; i < n; ++i) keep[i] = ops[i].y_dev != nullptr;
        seq_mark_resident(a.L, a.n, B, grid >> 3, nullptr, &keep);
    }
    if (n_fused_out) {
        *n_fused_out = 0;
        for (int i = 0; i < n; ++i) *n_fused_out += a.L[i].cfg == SEQ_CFG_C3C1_2ND;
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    // The outputs and the timed launches come from the instantiation the engine's own lists run (no stamp pointers: the stamps
    // live in the CLK build of the kernel); one more launch of the CLK build then writes the per-layer stamps (same results).
    SeqArgs a_run = a;
    a_run.clk = nullptr; a_run.clk2 = nullptr;
    if (launch_conv_seq(a_run, grid, s)) return fail(SMK_E_HIP, "conv_seq launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIPCHK(hipEventRecord(e0, s));
    for (int it = 1; it < iters; ++it)
        if (launch_conv_seq(a_run, grid, s)) return fail(SMK_E_HIP, "conv_seq launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIPCHK(hipEventRecord(e1, s));
    if (launch_conv_seq(a, grid, s)) return fail(SMK_E_HIP, "conv_seq launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    if (iters > 1) HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (usec_out) *usec_out = iters > 1 ? ms * 1000.f / (iters - 1) : 0.f;
    int e = 0;
    HIPCHK(hipMemcpy(&e, err, sizeof(int), hipMemcpyDeviceToHost));
    if (e) return fail(SMK_E_HIP, "conv_seq_kernel reported %s", e == 1 ? "an uneven distribution of workgroups over the XCDs" : "a team-barrier time-out");
    {
        unsigned long long h[2 * SEQ_MAX + 1], h2[(12 + 32) * SEQ_MAX];
        HIPCHK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        if (clk_us_out)
            for (int i = 0; i < n; ++i) {
                clk_us_out[2 * i] = (float)((h[1 + 2 * i] - h[2 * i]) / 100.0);
                clk_us_out[2 * i + 1] = (float)((h[2 + 2 * i] - h[1 + 2 * i]) / 100.0);
            }
        if (ck) {
            if (want2) HIPCHK(hipMemcpy(h2, clk2, sizeof(h2), hipMemcpyDeviceToHost));
            seq_print_clk(a, ids, "smk_op_conv_seq", h, want2 ? h2 : nullptr);
        }
    }
    for (int i = 0; i < n; ++i)
        if (ops[i].y_dev) {
            CvtOutParams co{outs[i].p, ops[i].y_dev, B, couts[i], outs[i].H, outs[i].W, outs[i].C, 0};
            if (launch_cvt_out(co, dtype, s)) return fail(SMK_E_HIP, "cvt_out launch failed");
        }
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int smk_op_conv2d(int dtype, int algo, const float *x_dev, int B, int Cin, int H, int W, const float *w_host,
                  const float *b_host, int Cout, int k, int stride, int pad, int dil, int relu,
                  const float *res_dev, float *y_dev, void *stream) {
    smk_conv_geom g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.k = k; g.stride = stride; g.pad = pad; g.dil = dil;
    g.relu = relu; g.res_mode = res_dev ? RES_PRE_RELU : RES_NONE;
    return smk_op_conv2d_ex(dtype, algo, &g, x_dev, w_host, b_host, res_dev, nullptr, y_dev, stream);
}

int smk_op_dw_xcorr(int dtype, const float *x_dev, const float *k_dev, int B, int C, int H, int W, int kh, int kw,
                    float *y_dev, void *stream) {
    if (!x_dev || !k_dev || !y_dev) return fail(SMK_E_ARG, "smk_op_dw_xcorr: null argument");
    if (C % 64 != 0) return fail(SMK_E_ARG, "smk_op_dw_xcorr: C must be a multiple of 64");
    if (kh > 5 || kw > 5 || kh > H || kw > W || (W - kw + 1 + 1) / 2 > 13)
        return fail(SMK_E_ARG, "smk_op_dw_xcorr: unsupported geometry");
    hipStream_t s = (hipStream_t)stream;
    TmpBufs tmp;
    const size_t es = esize(dtype);
    const int Ho = H - kh + 1, Wo = W - kw + 1;
    void *x, *k, *y;
    CHK(tmp.alloc(&x, (size_t)B * H * W * C * es));
    CHK(tmp.alloc(&k, (size_t)B * kh * kw * C * es));
    CHK(tmp.alloc(&y, (size_t)B * Ho * Wo * C * es));
    CvtInParams cx{x_dev, x, B, C, H, W, C, 0}, ck{k_dev, k, B, C, kh, kw, C, 0};
    if (launch_cvt_in(cx, dtype, s) || launch_cvt_in(ck, dtype, s)) return fail(SMK_E_HIP, "cvt_in launch failed");
    XcorrParams xp{x, k, y, B, H, W, kh, kw, Ho, Wo, C, C};
    if (launch_xcorr(xp, dtype, s)) return fail(SMK_E_HIP, "xcorr launch failed");
    CvtOutParams co{y, y_dev, B, C, Ho, Wo, C, 0};
    if (launch_cvt_out(co, dtype, s)) return fail(SMK_E_HIP, "cvt_out launch failed");
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int smk_op_maxpool3x3s2(int dtype, const float *x_dev, int B, int C, int H, int W, float *y_dev, void *stream) {
    if (!x_dev || !y_dev) return fail(SMK_E_ARG, "smk_op_maxpool3x3s2: null argument");
    if (C % 8 != 0) return fail(SMK_E_ARG, "smk_op_maxpool3x3s2: C must be a multiple of 8");
    hipStream_t s = (hipStream_t)stream;
    TmpBufs tmp;
    const size_t es = esize(dtype);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    void *x, *y;
    CHK(tmp.alloc(&x, (size_t)B * H * W * C * es));
    CHK(tmp.alloc(&y, (size_t)B * Ho * Wo * C * es));
    CvtInParams cx{x_dev, x, B, C, H, W, C, 0};
    if (launch_cvt_in(cx, dtype, s)) return fail(SMK_E_HIP, "cvt_in launch failed");
    PoolParams pp{x, y, B, H, W, C, Ho, Wo};
    if (launch_maxpool(pp, dtype, s)) return fail(SMK_E_HIP, "maxpool launch failed");
    CvtOutParams co{y, y_dev, B, C, Ho, Wo, C, 0};
    if (launch_cvt_out(co, dtype, s)) return fail(SMK_E_HIP, "cvt_out launch failed");
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ---- image ops either side of the network ------------------------------
int smk_crop_resize(const uint8_t *frames_dev, int64_t frame_stride_bytes, int H, int W, const int32_t *boxes,
                    const uint8_t *avg_bgr, int B, int model_sz, float *out_dev, void *stream) {
    if (!frames_dev || !boxes || !avg_bgr || !out_dev) return fail(SMK_E_ARG, "smk_crop_resize: null argument");
    if (H < 1 || W < 1 || model_sz < 1 || B < 1) return fail(SMK_E_ARG, "smk_crop_resize: bad geometry");
    for (int b0 = 0; b0 < B; b0 += CROP_MAX_B) {
        const int nb = B - b0 < CROP_MAX_B ? B - b0 : CROP_MAX_B;
        CropParams p;
        memset(&p, 0, sizeof(p));
        p.frames = frames_dev + (size_t)b0 * frame_stride_bytes;
        p.frame_stride = frame_stride_bytes;
        p.out = out_dev + (size_t)b0 * 3 * model_sz * model_sz;
        p.H = H; p.W = W; p.model_sz = model_sz;
        for (int i = 0; i < nb; ++i) {
            const int32_t *bx = boxes + 3 * (b0 + i);
            if (bx[2] < 1 || bx[2] > 32768) return fail(SMK_E_ARG, "smk_crop_resize: window size %d out of range", bx[2]);
            p.box[i][0] = bx[0]; p.box[i][1] = bx[1]; p.box[i][2] = bx[2];
            for (int k = 0; k < 3; ++k) p.avg[i][k] = avg_bgr[3 * (b0 + i) + k];
        }
        if (launch_crop_resize(p, nb, stream)) return fail(SMK_E_HIP, "crop_resize launch failed");
    }
    return 0;
}

int smk_paste_mask(const float *logits_dev, int mask_size, const double *inv_map, int B, int W, int H, float seg_thr,
                   float border, uint8_t *mask_out_dev, float *prob_out_dev, void *stream) {
    if (!logits_dev || !inv_map || (!mask_out_dev && !prob_out_dev)) return fail(SMK_E_ARG, "smk_paste_mask: null argument");
    if (W < 1 || H < 1 || mask_size < 1 || B < 1) return fail(SMK_E_ARG, "smk_paste_mask: bad geometry");
    for (int b0 = 0; b0 < B; b0 += CROP_MAX_B) {
        const int nb = B - b0 < CROP_MAX_B ? B - b0 : CROP_MAX_B;
        PasteParams p;
        memset(&p, 0, sizeof(p));
        p.logits = logits_dev + (size_t)b0 * mask_size * mask_size;
        p.mask_out = mask_out_dev ? mask_out_dev + (size_t)b0 * H * W : nullptr;
        p.prob_out = prob_out_dev ? prob_out_dev + (size_t)b0 * H * W : nullptr;
        p.ms = mask_size; p.W = W; p.H = H; p.seg_thr = seg_thr; p.border = border;
        for (int i = 0; i < nb; ++i)
            for (int k = 0; k < 6; ++k) p.inv_map[i][k] = inv_map[6 * (b0 + i) + k];
        if (launch_paste_mask(p, nb, stream)) return fail(SMK_E_HIP, "paste_mask launch failed");
    }
    return 0;
}

int smk_paste_labels(const float *logits_dev, int mask_size, const double *inv_map, int n_obj, int W, int H,
                     float seg_thr, float border, uint8_t *labels_out_dev, void *stream) {
    if (!logits_dev || !inv_map || !labels_out_dev) return fail(SMK_E_ARG, "smk_paste_labels: null argument");
    if (W < 1 || H < 1 || mask_size < 1 || n_obj < 1 || n_obj > CROP_MAX_B)
        return fail(SMK_E_ARG, "smk_paste_labels: bad geometry (1..%d objects)", CROP_MAX_B);
    PasteParams p;
    memset(&p, 0, sizeof(p));
    p.logits = logits_dev; p.mask_out = labels_out_dev;
    p.ms = mask_size; p.W = W; p.H = H; p.seg_thr = seg_thr; p.border = border;
    for (int i = 0; i < n_obj; ++i)
        for (int k = 0; k < 6; ++k) p.inv_map[i][k] = inv_map[6 * i + k];
    if (launch_paste_labels(p, n_obj, stream)) return fail(SMK_E_HIP, "paste_labels launch failed");
    return 0;
}

// time repeated launches of one conv geometry (operands are pseudo-random, not zeros: DVFS hygiene)
int smk_bench_conv(int dtype, int algo, const smk_conv_geom *g, int with_res, int iters, float *usec_out,
                   void *stream) {
    if (!g || !usec_out || iters < 1) return fail(SMK_E_ARG, "smk_bench_conv: bad argument");
    if (dtype != DT_F32 && dtype != DT_F16) return fail(SMK_E_ARG, "bad dtype");
    hipStream_t s = (hipStream_t)stream;
    PackedConv pc; Act in; ConvOpt o; int Ho, Wo;
    CHK(fill_geom(g, pc, in, o, Ho, Wo));
    const size_t es = esize(dtype);
    TmpBufs tmp;
    // SMK_BENCH_FILL (measurement aid): what the ACTIVATIONS hold -- uniform [-1, 1) (default), "zero", "relu" (negative values -> 0).  The
    // matrix pipes' clock under load depends on the operand bits (MI355X_MICROARCH.md: zero-filled operands +15..21 % TF/s).
    const char *fm = getenv("SMK_BENCH_FILL");
    const int fill_mode = !fm ? 0 : (!strcmp(fm, "zero") ? 1 : (!strcmp(fm, "relu") ? 2 : 0));
    auto fill = [&](void **dst, size_t elems, float scale, unsigned seed) -> int {
        CHK(tmp.alloc(dst, elems * es));
        std::vector<unsigned char> h(elems * es);
        unsigned st = seed * 2654435761u + 12345u;
        for (size_t i = 0; i < elems; ++i) {
            st = st * 1664525u + 1013904223u;
            float v = ((int)((st >> 9) & 0xffff) - 32768) * (scale / 32768.f);
            if (seed == 1 && fill_mode == 1) v = 0.f;                       // SMK_BENCH_FILL=zero: activations all zero
            if (seed == 1 && fill_mode == 2) v = v > 0.f ? v : 0.f;         // SMK_BENCH_FILL=relu: half of them zero, like a ReLU output
            if (dtype == DT_F16) ((_Float16 *)h.data())[i] = (_Float16)v; else ((float *)h.data())[i] = v;
        }
        HIPCHK(hipMemcpy(*dst, h.data(), elems * es, hipMemcpyHostToDevice));
        return 0;
    };
    CHK(fill(&in.p, (size_t)g->B * g->H * g->W * in.C, 1.0f, 1));
    CHK(fill(&pc.w, (size_t)pc.rows * pc.Kpad, 0.05f, 2));
    CHK(tmp.alloc((void **)&pc.bias, (size_t)pc.rows * 4));
    int *pos_dev = nullptr;
    if (g->pos_mul) {
        CHK(tmp.alloc((void **)&pos_dev, sizeof(int) * 2 * g->B));
        std::vector<int> ph(2 * g->B);
        for (int i = 0; i < 2 * g->B; ++i) ph[i] = 8 + (i * 5) % 9;
        HIPCHK(hipMemcpy(pos_dev, ph.data(), ph.size() * 4, hipMemcpyHostToDevice));
        o.pos = pos_dev;
    }
    const int mode = algo & 0xff;
    o.tile_code = (algo >> 8) & 0xff;
    Act out, res;
    out.H = Ho; out.W = Wo; out.C = rup(g->Cout, 8);
    float *nchw = nullptr;
    if (mode == 2) {
        CHK(tmp.alloc((void **)&nchw, (size_t)g->B * g->Cout * Ho * Wo * 4));
        o.nchw_out = nchw;
    } else {
        CHK(tmp.alloc(&out.p, (size_t)g->B * Ho * Wo * out.C * es));
        if (with_res) {
            res = out;
            CHK(fill(&res.p, (size_t)g->B * Ho * Wo * out.C, 1.0f, 3));
            o.res = &res; o.res_mode = RES_PRE_RELU;
        }
    }
    smk_ctx fake;
    fake.dtype = dtype;
    HIPCHK(hipGetDevice(&fake.device));
    CHK(op_ks_scratch(fake));
    ConvParams p;
    CHK(conv_params(&fake, pc, in, mode == 2 ? nullptr : &out, g->B, o, p));
    const TileChoice t = tile_from_code(o.tile_code, p, dtype);
    const int halo_bm = mode == 4 ? ((o.tile_code & 15) == 1 ? 128 : 64) : 0;     // timing only: K order is irrelevant
    const int wr = mode == 5 ? (o.tile_code & 15) : 0;
    if (mode == 5) {
        if (wr < 1 || wr > 8) return fail(SMK_E_ARG, "smk_bench_conv: wreg tile code 1..8");
        p.wgt_frag = p.wgt;                              // timing only: the fragment order is irrelevant
        if (!conv_wreg_eligible(p, dtype)) return fail(SMK_E_ARG, "smk_bench_conv: not eligible for conv_wreg_kernel");
    }
    const int wr_stages = wreg_stages_from_code(o.tile_code);
    if (mode == 6 && !conv_pp_eligible(p, dtype)) return fail(SMK_E_ARG, "smk_bench_conv: not eligible for conv_pp_kernel");
    auto launch = [&]() {
        if (mode == 6) return launch_conv_pp(p, s);
        if (wr) {
            ConvBatch cb;
            cb.n = 1;
            cb.p[0] = p;
            return launch_conv_wreg_batch(cb, WREG_TILE[wr][0], WREG_TILE[wr][1], wr_stages, s);
        }
        return halo_bm ? launch_conv_halo(p, dtype, halo_bm, s) : launch_conv_mfma(p, dtype, t, s);
    };
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (launch()) return fail(SMK_E_HIP, "conv launch failed or geometry not eligible");
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (launch()) return fail(SMK_E_HIP, "conv launch failed");
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *usec_out = ms * 1000.f / iters;
    return 0;
}

// host-only walk of the packed matrix + the kernels' gather function (no GPU needed)
int smk_host_conv2d_ex(const smk_conv_geom *g, const float *x, const float *w, const float *b, const float *res,
                       const int32_t *pos, float *y) {
    if (!g || !x || !w || !y) return fail(SMK_E_ARG, "smk_host_conv2d_ex: null argument");
    PackedConv pc; Act in; ConvOpt o; int Ho, Wo;
    CHK(fill_geom(g, pc, in, o, Ho, Wo));
    std::vector<float> rows, bias;
    pack_host(g, pc, w, b, rows, bias);
    // NCHW -> NHWC (channel padded) on the host, mirroring cvt_in_kernel
    std::vector<float> xin((size_t)g->B * g->H * g->W * in.C, 0.f);
    for (int bb = 0; bb < g->B; ++bb)
        for (int c = 0; c < g->Cin; ++c)
            for (int yy = 0; yy < g->H; ++yy)
                for (int xx = 0; xx < g->W; ++xx)
                    xin[(((size_t)bb * g->H + yy) * g->W + xx) * in.C + c] =
                        x[(((size_t)bb * g->Cin + c) * g->H + yy) * g->W + xx];
    in.p = xin.data();
    pc.w = rows.data();
    pc.bias = bias.data();
    o.pos = pos;
    Act out;
    out.H = Ho; out.W = Wo; out.C = rup(g->Cout, 8);
    std::vector<float> obuf((size_t)g->B * Ho * Wo * out.C, 0.f);
    out.p = obuf.data();
    smk_ctx fake;
    fake.device = -1;
    ConvParams p;
    CHK(conv_params(&fake, pc, in, &out, g->B, o, p));
    for (int m = 0; m < p.M; ++m) {
        const RowInfo r = row_info(p, m, p.pos);
        for (int n = 0; n < g->Cout; ++n) {
            double acc = 0.0;
            for (int kvec = 0; kvec < p.K; kvec += 4) {
                const KDecode d = decode_k(kvec, p.Ci, p.kw);
                const long off = gather_offset(p, r, d, p.cin_off);
                if (off < 0) continue;
                for (int e = 0; e < 4; ++e) acc += (double)xin[off + e] * (double)rows[(size_t)n * p.Kpad + kvec + e];
            }
            double v = acc + bias[n];
            const int hw = Ho * Wo, bb = m / hw, ps = m - bb * hw;
            const double rv = (res && g->res_mode) ? res[((size_t)bb * g->Cout + n) * hw + ps] : 0.0;
            if (g->res_mode == RES_PRE_RELU) v += rv;
            if (g->relu) v = v > 0 ? v : 0;
            if (g->res_mode == RES_POST_RELU) v += rv;
            y[((size_t)bb * g->Cout + n) * hw + ps] = (float)v;
        }
    }
    return 0;
}

// Which kernel and workgroup shape does the engine pick for one convolution of this geometry?  Host only (CPU tests pin the
// measured layer rules with it).  Mirrors the decision order of run_conv: register-fed kernel (wreg_choice), else the
// patch-sharing kernel (halo_choice), else the generic one (tile_from_code); *seq_cfg = the conv_seq_kernel tile code the layer
// gets inside a persistent sequence, or -1 when it cannot be part of one.
int smk_host_plan_conv(const smk_conv_geom *g, int dtype, int with_res, int *kernel, int *bm, int *bn, int *seq_cfg) {
    if (!g || !kernel || !bm || !bn || !seq_cfg) return fail(SMK_E_ARG, "smk_host_plan_conv: null argument");
    if (dtype != DT_F32 && dtype != DT_F16) return fail(SMK_E_ARG, "smk_host_plan_conv: dtype");
    PackedConv pc; Act in; ConvOpt o; int Ho, Wo;
    CHK(fill_geom(g, pc, in, o, Ho, Wo));
    static float dummy[64];
    in.p = dummy;
    pc.w = dummy; pc.bias = dummy;
    if (g->k == 3) pc.w_halo = dummy;
    if (dtype == DT_F16) pc.w_frag = dummy;
    if (dtype == DT_F16 && g->k == 3 && pc.Kpad == 9 * pc.Ci) pc.w_frag_halo = dummy;
    Act out, res;
    out.H = Ho; out.W = Wo; out.C = rup(g->Cout, 8); out.p = dummy;
    res = out;
    if (with_res) { o.res = &res; o.res_mode = RES_PRE_RELU; }
    smk_ctx fake;
    fake.device = -1;
    fake.dtype = dtype;
    ConvParams p;
    CHK(conv_params(&fake, pc, in, &out, g->B, o, p));
    const TileChoice t = tile_from_code(0, p, dtype);
    int hb = halo_choice(pc, p, o, dtype);
    if (hb && conv_ksplit(p, dtype, t) > 1) hb = 0;
    const int wr = (hb && g_tune.wreg < 2 && g_tune.wreg_policy == 0) ? 0 : wreg_choice(p, o, dtype);
    if (pp_choice(p, o, dtype)) { *kernel = 3; *bm = 256; *bn = 256; }
    else if (wr) { *kernel = 2; *bm = WREG_TILE[wr][0]; *bn = WREG_TILE[wr][1]; }
    else if (hb) { *kernel = 1; *bm = hb; *bn = 128; }
    else { *kernel = 0; *bm = t.bm; *bn = t.bn; }
    SeqLayer L;
    *seq_cfg = seq_layer_from(p, dtype, L) ? (int)L.cfg : -1;
    return 0;
}

}  // extern "C"
