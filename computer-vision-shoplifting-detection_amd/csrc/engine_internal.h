// Internals shared by the engine's translation units (engine_load / engine_memory / engine_plans / engine_run / engine_abi /
// engine_ops .hip): the .mi355w records, the engine object behind the opaque mi355_yolo handle and the functions that cross files.
// Replaces what the reference reaches through ultralytics (model.py:18,38): Model.__init__/AutoBackend (weights + fuse),
// BasePredictor.stream_inference (preprocess -> model -> postprocess).
#pragma once
#include "common.h"
#include "../../include/mi355_yolo.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

namespace mi355 {

extern thread_local std::string g_err;          // engine_abi.hip; mi355_last_error() reads it
inline int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(MI355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)
#define KCHK(expr)                                                                           \
    do {                                                                                     \
        const char* m_ = (expr);                                                             \
        if (m_) return fail(MI355_EHIP, std::string("launch failed: ") + m_);                \
    } while (0)

enum { OP_STEM = 0, OP_CONV = 1, OP_UPSAMPLE = 2, OP_SPPF_POOL = 3 };

#pragma pack(push, 1)
struct FileHeader {
    char magic[8];
    uint32_t version, header_bytes;
    uint32_t family, scale, task, nc, nkpt, kdim, reg_max;
    uint32_t n_buffers, n_ops, n_convs, n_levels;
    uint32_t json_off, json_bytes;
    uint64_t data_bytes;
};
struct FileBuf { uint32_t channels, stride_div; };
struct FileOp { int32_t type, k, s, act, src_buf, src_choff, src_c, dst_buf, dst_choff, dst_c, res_buf, res_choff, conv, pad, r0, r1; };
struct FileConv { char name[64]; uint32_t cin, cout, k, s, pad, act; uint64_t w_off, b_off; };
struct FileLevel { uint32_t buf, box_off, cls_off, kpt_off, stride; };
#pragma pack(pop)

struct DevConv { float* wpk = nullptr; float* bias = nullptr; float* w_raw = nullptr; void* w_frag = nullptr; };   // w_frag: stem3_weight_frags (half k3 stems)

// LetterBox geometry (data/augment.py:LetterBox, auto=True, scaleup=True, center=True, stride 32) and the
// scale-back constants of utils/ops.py:scale_boxes / scale_coords, in the same double arithmetic as Python.
struct Geometry {
    int h0, w0, Hl, Wl;          // original and letterboxed size
    int Hr, Wr, top, left;       // resized region
    bool resize, identity;
    double gain; double pad_x, pad_y, kpad_x, kpad_y;
};

inline double py_round(double x) { return std::nearbyint(x); }   // round-half-even, like Python's round()
inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

inline Geometry make_geometry(int h0, int w0, int imgsz) {
    Geometry g{};
    g.h0 = h0; g.w0 = w0;
    const double r = std::min((double)imgsz / h0, (double)imgsz / w0);
    g.Wr = (int)py_round(w0 * r); g.Hr = (int)py_round(h0 * r);
    double dw = imgsz - g.Wr, dh = imgsz - g.Hr;
    dw = std::fmod(dw, 32.0); dh = std::fmod(dh, 32.0);
    dw /= 2; dh /= 2;
    const int top = (int)py_round(dh - 0.1), bottom = (int)py_round(dh + 0.1);
    const int left = (int)py_round(dw - 0.1), right = (int)py_round(dw + 0.1);
    g.top = top; g.left = left;
    g.Hl = g.Hr + top + bottom; g.Wl = g.Wr + left + right;
    g.resize = (g.Wr != w0) || (g.Hr != h0);
    g.identity = !g.resize && top == 0 && left == 0 && bottom == 0 && right == 0;
    g.gain = std::min((double)g.Hl / h0, (double)g.Wl / w0);
    g.pad_x = py_round((g.Wl - w0 * g.gain) / 2 - 0.1);
    g.pad_y = py_round((g.Hl - h0 * g.gain) / 2 - 0.1);
    g.kpad_x = (g.Wl - w0 * g.gain) / 2;
    g.kpad_y = (g.Hl - h0 * g.gain) / 2;
    return g;
}

// cv2.resize(INTER_LINEAR) coefficient table: for each destination index: source index, 2 taps in 1/2048 units
inline void resize_table(int dn, int sn, std::vector<int>& tab) {
    tab.resize((size_t)dn * 3);
    const double scale = (double)sn / dn;
    for (int d = 0; d < dn; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(fx);
        fx -= (float)s;
        if (s < 0) { s = 0; fx = 0.f; }
        if (s >= sn - 1) { s = sn - 1; fx = 0.f; }
        tab[d * 3] = s;
        tab[d * 3 + 1] = (int)std::lrintf((1.f - fx) * 2048.f);
        tab[d * 3 + 2] = (int)std::lrintf(fx * 2048.f);
    }
}

}  // namespace mi355

using namespace mi355;

struct mi355_yolo {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // H2D of the next chunk overlaps the current chunk's kernels (host-frame entry point)
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    FileHeader hdr{};
    std::vector<FileBuf> bufs;
    std::vector<FileOp> ops;
    std::vector<FileConv> convs;
    std::vector<FileLevel> levels;
    std::vector<DevConv> dconv;
    float* lut = nullptr;
    float* zeros = nullptr;             // 256 zero bytes: DMA source of padded LDS slots
    int chunk = 64;
    // tuned launch-plan choice per (frames, H, W): candidate index per op, so a shape seen before is not re-timed
    std::vector<std::pair<std::array<int, 3>, std::vector<int>>> tuned;
    int autotune = 32;                  // candidate launch plans timed per conv when a shape is first seen (0 = off)
    long long n_params = 0, macs640 = 0;

    // per-shape state: activation buffers are GROW-ONLY for a letterboxed size (alloc_nb frames); launch plans belong to
    // (cur_nb, cur_H, cur_W) and are rebuilt -- from the tuned-choice cache, without touching the buffers -- when only the
    // frame count of a call changes (sweep tails, track() at n = 1 between batched predicts)
    int cur_nb = 0, cur_H = 0, cur_W = 0, alloc_nb = 0;
    long long act_bytes = 0;            // bytes of activation buffers currently allocated
    unsigned long long model_hash = 0;  // FNV-1a of the .mi355w image: key of the persisted plan choices
    std::vector<float*> dbuf;           // activation buffers (fp32, or fp16 bytes behind a float* when `half`): views into `arena`
    bool host_only = false;             // mi355_memory_plan: program analysis without a device (no weights uploaded, nothing launched)
    char* arena = nullptr;              // ONE allocation; buffers whose lifetimes cannot overlap under ANY legal schedule share bytes
    int mem_reuse = 1;                  // MI355_OPT_NO_MEM_REUSE: every buffer gets bytes of its own (the round-1/2 layout)
    // mi355_opts (include/mi355_yolo.h); the MI355_* environment variables override them for A/B runs only
    int opt_flags = 0;                  // MI355_OPT_*
    bool fast_act = false;              // opts.fast_act: v_exp / v_rcp SiLU in the fp32 conv epilogues (tolerance mode)
    std::string plan_dir;               // shipped plan files, read-only, looked up first ("" = none)
    std::string plan_cache_dir;         // freshly timed choices are written here ("" = not persisted)
    bool plan_cache_on = true;
    long long act_bytes_noreuse = 0;    // what the same shape takes without sharing (reported beside act_bytes)
    std::vector<std::vector<unsigned long long>> anc;   // anc[i] = bitset over ops: RAW ancestors of op i (transitive)
    std::vector<int> dbuf_cs;           // pixel stride in ELEMENTS of the buffer's dtype
    std::vector<int> dbuf_es;           // element size in bytes: 4, or 2 for the fp16 buffers of the half=True path
    bool half = false;                  // opts.half: fp16 storage of activations / weights, fp32 arithmetic (conv_igemm_f16.hip)
    float* view(int buf, int choff) const { return (float*)((char*)dbuf[buf] + (size_t)choff * dbuf_es[buf]); }
    std::vector<ConvLaunch> plans;      // per op (valid for OP_CONV)
    // Upsample -> Concat -> Conv1x1 of the neck, fused on the conv's read side: fuse_up[j] = index of the OP_UPSAMPLE op whose
    // output only conv op j reads (or -1); fused_away[i] = that upsample is not launched.  Decided when the weights are
    // loaded (program structure) and confirmed per shape (a v4 launch plan must exist), MI355_FUSE_UPSAMPLE=0 disables it.
    std::vector<int> fuse_up; std::vector<char> fused_away;
    // Conv3x3 -> Conv1x1 fused into one launch: fuse2[i] = index of the pointwise conv op whose ONLY input is conv op
    // i's output slice, which nobody else reads (or -1): the stride-2 convs in front of every C2f / C3 and the last two convs
    // of every head branch.  Decided from the program at load time, confirmed per shape (a fused launch plan must exist);
    // skip_op[j] = the pointwise op j runs inside its producer's launch.  MI355_FUSE_1X1=0 disables it.
    std::vector<int> fuse2; std::vector<char> skip_op;
    // ... and its generalisation to the tail of a C2f block: the LAST Bottleneck's second 3x3 conv (with its residual) feeds only
    // C2f.cv2, a pointwise conv over cat(ys) whose input slice ENDS with that conv's output: fuse2_lead[i] = the channels of the
    // concat buffer in front of it (read by the fused pointwise stage straight from global memory), 0 = exact-slice pairs.
    std::vector<int> fuse2_lead;
    // Small chunks leave most of the chip idle inside one conv launch, but the graph has independent branches (the box / class /
    // keypoint chains of the three head levels run beside the rest of the neck): ops are dealt to a few HIP streams along the
    // program's dependency DAG (RAW on buffer slices), in depth order, a chain inheriting its producer's stream; an op waits
    // on the events of producers that live on other streams.  Measured gain: +14 % at batch 1, +10 % at 8, +3 % at 64 and
    // still +2 % at 512 (the tails of one launch fill with the blocks of another); MI355_STREAMS=1 turns it off.
    int n_streams = 4, streams_max_batch = 1 << 30, streams_min_batch = 6;   // below 6 frames per pass one in-order stream is faster
                                                  // (round 2, merged + fused program: batch 1 1,990 vs 1,844 frames/s, batch 4 4,780 vs 4,530;
                                                  // batch 8 6,150 vs 6,360): the cross-stream event waits cost more than the overlap buys
    std::vector<hipStream_t> aux;             // streams 1 .. n_streams-1 (0 = `stream`)
    std::vector<hipEvent_t> op_done;          // per op: recorded after its launch when someone on another stream waits for it
    hipEvent_t ev_fork = nullptr;
    std::vector<int> sched_order, op_stream;  // launch order (depth, index) and stream of each op
    std::vector<std::vector<int>> op_xdeps;   // producers on other streams
    std::vector<char> op_signals;             // op has a consumer on another stream (or is a head output: decode joins on it)
    std::vector<int> leaf_ops;                // ops that write the head-level buffers
    std::vector<std::vector<int>> deps;       // RAW producers of every op (program order indices)
    // Grouped launches (conv_f32_group.hip), the single-stream regime's answer to the idle chip: the launched ops are list-
    // scheduled into STEPS (all ops of a step are mutually independent: every producer ran in an earlier step); the conv ops
    // of a step whose tuned kernel is on the group kernel's menu run as ONE grid when the stopwatch says that beats the
    // separate launches.  group_sel[i] >= 0: op i runs inside its step's group with plan group_sel[i] of its candidate list
    // (fused list when its pointwise consumer runs inside it).  MI355_GROUPS=0 turns it off.
    struct Step { std::vector<int> singles; int group = -1; };
    std::vector<Step> steps;
    std::vector<GroupLaunch> groups;
    std::vector<int> group_sel;
    int use_groups = 1, group_max_batch = 5;
    float* pred = nullptr; float2* best = nullptr; unsigned long long* keys = nullptr;
    int A = 0, Apow2 = 0;
    uint8_t* lbox = nullptr;            // letterboxed frames of one chunk (also the stable stem input of the graph path)
    std::vector<std::pair<int, hipGraphExec_t>> graphs;   // (frames in chunk, captured stem..decode sequence)
    int use_graph = 0;                  // MI355_GRAPH=1: replay stem..decode as a hipGraph (measured: no gain, the small-batch
                                        // regime is bound by per-kernel latency of tiny grids, not by host launches)
    // per-call scratch (grown on demand)
    uint8_t* d_in = nullptr; size_t d_in_bytes = 0;
    mi355_det* d_rows = nullptr; int* d_counts = nullptr; size_t rows_cap = 0; int counts_cap = 0;
    mi355_det* h_rows = nullptr; int* h_counts = nullptr; size_t h_rows_cap = 0; int h_counts_cap = 0;
    mi355_det* d_packed = nullptr; int* d_offsets = nullptr; size_t packed_cap = 0; int offsets_cap = 0;   // rows compacted on the GPU before the D2H copy
    unsigned* d_cmask = nullptr; unsigned* h_cmask = nullptr; int cmask_words = 0;
    int* d_xtab = nullptr; int* d_ytab = nullptr; int tab_h0 = -1, tab_w0 = -1, tab_imgsz = -1;
    float* d_rawhead = nullptr; size_t rawhead_floats = 0;
    unsigned long long plan_hash = 0;   // fingerprint of the candidate lists + the chosen indices of the current shape
    int plan_source = 0, plan_launches = 0;   // 0 static guess (autotune off), 1 memory, 2 this machine's plan cache, 3 tuned now, 4 shipped plan file; launches of one pass (stem..last conv)
    bool async_pending = false;         // mi355_yolo_infer_device_async work may still be in flight on `stream`
    // timing
    bool profiling = false;
    mi355_timing last{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> pev;        // profiling events

    int no() const { return 4 + (int)hdr.nc + (int)(hdr.nkpt * hdr.kdim); }
    void free_shape();
    ~mi355_yolo();
};

namespace mi355 {

// event bookkeeping for per-kind timing
enum Kind { K_LETTERBOX, K_STEM, K_CONV, K_POOL, K_UPSAMPLE, K_DECODE, K_NMS, K_COUNT };
struct Prof {
    mi355_yolo* h; size_t used = 0; std::vector<std::pair<int, size_t>> spans;
    int begin(int kind) {
        if (!h->profiling) return 0;
        if (used + 2 > h->pev.size()) { for (int i = 0; i < 64; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return -1; h->pev.push_back(e); } }
        spans.push_back({kind, used});
        return hipEventRecord(h->pev[used], h->stream) == hipSuccess ? 0 : -1;
    }
    int end() {
        if (!h->profiling) return 0;
        const int r = hipEventRecord(h->pev[used + 1], h->stream) == hipSuccess ? 0 : -1;
        used += 2; return r;
    }
};

struct DevMem {   // RAII for the one-shot operator entry points
    std::vector<void*> ptrs;
    ~DevMem() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) ptrs.push_back(*p); return e; }
};

// engine_load.hip: .mi355w image -> op program, weights on the device, dependency DAG + stream assignment
int parse_blob(mi355_yolo* h, const uint8_t* blob, size_t n);
int create_impl(const uint8_t* blob, size_t nbytes, int device_id, const mi355_opts* opts, mi355_yolo** out);
// engine_memory.hip: liveness-based placement of the activation buffers in ONE arena (host arithmetic only)
void plan_memory(mi355_yolo* h, int nb, int Hl, int Wl, std::vector<size_t>* off_out, std::vector<size_t>* bytes_out,
                 size_t* arena_out, size_t* plain_out);
// engine_plans.hip: buffers + launch plans of (frames per pass, letterboxed H, W): candidate lists, plan files, the stopwatch
int ensure_shape(mi355_yolo* h, int nb, int Hl, int Wl);
// engine_run.hip: one pass of the net, one chunk, one call
int launch_net(mi355_yolo* h, Prof& pf, const uint8_t* stem_in, int nb, const Geometry& g, bool full_pred);
int run_chunk(mi355_yolo* h, Prof& pf, const uint8_t* frames_dev, int nb, const Geometry& g, bool full_pred);
int prepare_geometry(mi355_yolo* h, const Geometry& g, int imgsz);
int infer_impl(mi355_yolo* h, const uint8_t* src, bool src_on_device, int n, int height, int width, int row_stride,
               float conf, float iou, const int* classes, int n_classes, int max_det, int imgsz,
               mi355_det* out_rows, int cap, int* out_counts, mi355_det* dev_rows = nullptr, int* dev_counts = nullptr,
               int* dev_total = nullptr);

}  // namespace mi355
