// Host side of libmi355yolo.so: .mi355w reader, memory plan, launch sequence and the C ABI of
// include/mi355_yolo.h.  Replaces what the reference reaches through ultralytics (model.py:18,38):
// Model.__init__/AutoBackend (weights + fuse), BasePredictor.stream_inference (preprocess -> model -> postprocess).
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

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

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

struct DevConv { float* wpk = nullptr; float* bias = nullptr; float* w_raw = nullptr; };

// LetterBox geometry (data/augment.py:LetterBox, auto=True, scaleup=True, center=True, stride 32) and the
// scale-back constants of utils/ops.py:scale_boxes / scale_coords, in the same double arithmetic as Python.
struct Geometry {
    int h0, w0, Hl, Wl;          // original and letterboxed size
    int Hr, Wr, top, left;       // resized region
    bool resize, identity;
    double gain; double pad_x, pad_y, kpad_x, kpad_y;
};

static double py_round(double x) { return std::nearbyint(x); }   // round-half-even, like Python's round()
static size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

static Geometry make_geometry(int h0, int w0, int imgsz) {
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
static void resize_table(int dn, int sn, std::vector<int>& tab) {
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
    int mem_reuse = 1;                  // MI355_MEM_REUSE=0: every buffer gets bytes of its own (the round-1/2 layout)
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
    int plan_source = 0, plan_launches = 0;   // 0 static guess (autotune off), 1 memory, 2 plan file, 3 tuned now; launches of one pass (stem..last conv)
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

void mi355_yolo::free_shape() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    graphs.clear();
    if (arena) (void)hipFree(arena);
    arena = nullptr;
    dbuf.clear(); dbuf_cs.clear(); dbuf_es.clear(); plans.clear();
    groups.clear(); steps.clear();
    if (pred) (void)hipFree(pred); if (best) (void)hipFree(best); if (keys) (void)hipFree(keys);
    if (lbox) (void)hipFree(lbox);
    pred = nullptr; best = nullptr; keys = nullptr; lbox = nullptr;
    cur_nb = cur_H = cur_W = alloc_nb = 0;
    act_bytes = 0;
}

mi355_yolo::~mi355_yolo() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);      // an asynchronous call may still be running on the buffers freed below
    for (auto st : aux) if (st) (void)hipStreamSynchronize(st);
    free_shape();
    for (auto& c : dconv) { if (c.wpk) (void)hipFree(c.wpk); if (c.bias) (void)hipFree(c.bias); if (c.w_raw) (void)hipFree(c.w_raw); }
    if (lut) (void)hipFree(lut);
    if (zeros) (void)hipFree(zeros);
    if (d_in) (void)hipFree(d_in);
    if (d_rows) (void)hipFree(d_rows); if (d_counts) (void)hipFree(d_counts);
    if (d_packed) (void)hipFree(d_packed); if (d_offsets) (void)hipFree(d_offsets);
    if (h_rows) (void)hipHostFree(h_rows); if (h_counts) (void)hipHostFree(h_counts);
    if (d_cmask) (void)hipFree(d_cmask); if (h_cmask) (void)hipHostFree(h_cmask);
    if (d_xtab) (void)hipFree(d_xtab); if (d_ytab) (void)hipFree(d_ytab);
    if (d_rawhead) (void)hipFree(d_rawhead);
    if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1);
    for (auto e : pev) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) { if (ev_copied[i]) (void)hipEventDestroy(ev_copied[i]); if (ev_consumed[i]) (void)hipEventDestroy(ev_consumed[i]); }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    for (auto e : op_done) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto st : aux) if (st) (void)hipStreamDestroy(st);
    if (stream) (void)hipStreamDestroy(stream);
}

namespace mi355 {

static int build_schedule(mi355_yolo* h);

static int parse_blob(mi355_yolo* h, const uint8_t* blob, size_t n) {
    if (n < sizeof(FileHeader) || std::memcmp(blob, "MI355YW1", 8) != 0) return fail(MI355_EFORMAT, "not a .mi355w file (bad magic)");
    std::memcpy(&h->hdr, blob, sizeof(FileHeader));
    const FileHeader& H = h->hdr;
    if (H.version != 1) return fail(MI355_EFORMAT, "unsupported .mi355w version");
    size_t p = sizeof(FileHeader);
    const size_t need = p + sizeof(FileBuf) * H.n_buffers + sizeof(FileOp) * H.n_ops + sizeof(FileConv) * H.n_convs +
                        sizeof(FileLevel) * H.n_levels;
    if (need > n || (size_t)H.header_bytes + H.data_bytes > n || H.n_levels > 4 || H.n_levels < 1)
        return fail(MI355_EFORMAT, ".mi355w file is truncated or inconsistent");
    h->bufs.resize(H.n_buffers); std::memcpy(h->bufs.data(), blob + p, sizeof(FileBuf) * H.n_buffers); p += sizeof(FileBuf) * H.n_buffers;
    h->ops.resize(H.n_ops);      std::memcpy(h->ops.data(), blob + p, sizeof(FileOp) * H.n_ops);       p += sizeof(FileOp) * H.n_ops;
    h->convs.resize(H.n_convs);  std::memcpy(h->convs.data(), blob + p, sizeof(FileConv) * H.n_convs); p += sizeof(FileConv) * H.n_convs;
    h->levels.resize(H.n_levels); std::memcpy(h->levels.data(), blob + p, sizeof(FileLevel) * H.n_levels);
    if (H.nkpt * H.kdim > MI355_MAX_KPT_FLOATS) return fail(MI355_EFORMAT, "keypoint shape larger than 17x3 is not supported");
    {
        unsigned long long hsh = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) { hsh ^= blob[i]; hsh *= 1099511628211ull; }
        h->model_hash = hsh;
    }
    // validate the program
    for (const FileOp& o : h->ops) {
        auto okv = [&](int b, int off, int c) { return b >= 0 && b < (int)H.n_buffers && off >= 0 && (off & 3) == 0 && off + c <= (int)h->bufs[b].channels; };
        if (o.type != OP_STEM && !okv(o.src_buf, o.src_choff, o.src_c)) return fail(MI355_EFORMAT, "op reads outside its buffer");
        if (!okv(o.dst_buf, o.dst_choff, o.type == OP_SPPF_POOL ? 3 * o.src_c : o.dst_c)) return fail(MI355_EFORMAT, "op writes outside its buffer");
        if (o.res_buf >= 0 && !okv(o.res_buf, o.res_choff, o.dst_c)) return fail(MI355_EFORMAT, "residual outside its buffer");
        if ((o.type == OP_STEM || o.type == OP_CONV) && (o.conv < 0 || o.conv >= (int)H.n_convs)) return fail(MI355_EFORMAT, "bad conv index");
    }
    {
        // The multi-stream schedule orders ops by read-after-write only.  That is complete iff every channel of a buffer has
        // ONE writer and every reader comes after it in program order (no WAW, no WAR): checked here, so that a program
        // that recycles buffer slices is refused instead of racing silently across streams.
        auto written = [&](const FileOp& o) { return o.type == OP_SPPF_POOL ? 3 * o.src_c : o.dst_c; };
        auto overlaps = [](int a0, int ac, int b0, int bc) { return a0 < b0 + bc && b0 < a0 + ac; };
        for (size_t i = 0; i < h->ops.size(); ++i) {
            const FileOp& w = h->ops[i];
            for (size_t j = 0; j < h->ops.size(); ++j) {
                const FileOp& o = h->ops[j];
                if (j > i && o.dst_buf == w.dst_buf && overlaps(o.dst_choff, written(o), w.dst_choff, written(w)))
                    return fail(MI355_EFORMAT, "program writes a buffer slice twice (buffer reuse is not supported)");
                if (j < i) {
                    const bool rd = o.type != OP_STEM && o.src_buf == w.dst_buf && overlaps(o.src_choff, o.src_c, w.dst_choff, written(w));
                    const bool rr = o.res_buf == w.dst_buf && overlaps(o.res_choff, o.dst_c, w.dst_choff, written(w));
                    if (rd || rr) return fail(MI355_EFORMAT, "program reads a buffer slice before the op that writes it");
                }
            }
        }
    }
    h->n_params = H.reg_max; h->macs640 = 0;
    for (const FileOp& o : h->ops) {
        if (o.type != OP_STEM && o.type != OP_CONV) continue;
        const FileConv& c = h->convs[o.conv];
        const long long sd = h->bufs[o.dst_buf].stride_div;
        h->n_params += (long long)c.cout * c.cin * c.k * c.k + c.cout;
        h->macs640 += (long long)c.cout * c.cin * c.k * c.k * (640 / sd) * (640 / sd);
    }
    // upload weights
    const uint8_t* data = blob + H.header_bytes;
    h->dconv.resize(H.n_convs);
    std::vector<float> tmp;
    for (size_t i = 0; i < h->convs.size(); ++i) {
        const FileConv& c = h->convs[i];
        const size_t wn = (size_t)c.cout * c.cin * c.k * c.k;
        if (c.w_off + wn * 4 > H.data_bytes || c.b_off + (size_t)c.cout * 4 > H.data_bytes) return fail(MI355_EFORMAT, "tensor outside the data region");
        const float* w = (const float*)(data + c.w_off);
        const float* b = (const float*)(data + c.b_off);
        if (h->host_only) continue;
        DevConv& d = h->dconv[i];
        const int bn = round_up((int)c.cout, 16);
        tmp.assign(bn, 0.f);
        std::memcpy(tmp.data(), b, (size_t)c.cout * 4);
        HIPCHK(hipMalloc(&d.bias, bn * 4));
        HIPCHK(hipMemcpy(d.bias, tmp.data(), bn * 4, hipMemcpyHostToDevice));
        if (c.cin == 3) {                      // stem: raw OIHW, read by stem_mfma_u8
            HIPCHK(hipMalloc(&d.w_raw, wn * 4));
            HIPCHK(hipMemcpy(d.w_raw, w, wn * 4, hipMemcpyHostToDevice));
        } else if (h->half) {
            const size_t pn = packed_weight_halfs(c.cout, c.cin, c.k);
            std::vector<uint16_t> th(pn);
            pack_conv_weights_f16(w, c.cout, c.cin, c.k, th.data());
            HIPCHK(hipMalloc(&d.wpk, pn * 2));
            HIPCHK(hipMemcpy(d.wpk, th.data(), pn * 2, hipMemcpyHostToDevice));
        } else {
            const size_t pn = packed_weight_floats(c.cout, c.cin, c.k);
            tmp.resize(pn);
            pack_conv_weights(w, c.cout, c.cin, c.k, tmp.data());
            HIPCHK(hipMalloc(&d.wpk, pn * 4));
            HIPCHK(hipMemcpy(d.wpk, tmp.data(), pn * 4, hipMemcpyHostToDevice));
        }
    }
    // fusable upsamples: written slice (D, o, C) read by exactly one later op, a pointwise conv whose input view starts at o
    h->fuse_up.assign(h->ops.size(), -1); h->fused_away.assign(h->ops.size(), 0);
    const char* fenv = getenv("MI355_FUSE_UPSAMPLE");
    if (!fenv || atoi(fenv) != 0) {
        for (size_t i = 0; i < h->ops.size(); ++i) {
            const FileOp& u = h->ops[i];
            if (u.type != OP_UPSAMPLE) continue;
            int reader = -1, n_readers = 0;
            for (size_t j = 0; j < h->ops.size(); ++j) {
                const FileOp& o = h->ops[j];
                if (j == i || o.type == OP_STEM) continue;
                const bool reads = o.src_buf == u.dst_buf && o.src_choff < u.dst_choff + u.src_c && u.dst_choff < o.src_choff + o.src_c;
                const bool reads_res = o.res_buf == u.dst_buf && o.res_choff < u.dst_choff + u.src_c && u.dst_choff < o.res_choff + o.dst_c;
                if (reads || reads_res) { ++n_readers; reader = reads && !reads_res ? (int)j : -2; }
            }
            bool is_head = false;
            for (const FileLevel& lv : h->levels) is_head |= ((int)lv.buf == u.dst_buf);
            if (n_readers != 1 || reader < 0 || reader < (int)i || is_head) continue;
            const FileOp& c = h->ops[reader];
            if (c.type != OP_CONV || h->convs[c.conv].k != 1 || h->convs[c.conv].s != 1) continue;
            if (c.src_choff != u.dst_choff || c.src_c < u.src_c) continue;      // the upsampled operand must lead the conv's input
            h->fuse_up[reader] = (int)i; h->fused_away[i] = 1;
        }
    }
    h->fuse2.assign(h->ops.size(), -1); h->skip_op.assign(h->ops.size(), 0); h->fuse2_lead.assign(h->ops.size(), 0);
    const char* f2env = getenv("MI355_FUSE_1X1");
    const char* f3env = getenv("MI355_FUSE_TAIL");
    const bool fuse_tail = (!f3env || atoi(f3env) != 0) && !h->half;      // fp32 kernels only
    if (!f2env || atoi(f2env) != 0) {
        for (size_t i = 0; i < h->ops.size(); ++i) {
            const FileOp& p3 = h->ops[i];
            if (p3.type != OP_CONV || h->convs[p3.conv].k != 3 || (p3.res_buf >= 0 && !fuse_tail)) continue;
            bool is_head = false;
            for (const FileLevel& lv : h->levels) is_head |= ((int)lv.buf == p3.dst_buf);
            if (is_head) continue;
            int reader = -1, n_readers = 0;
            for (size_t j = 0; j < h->ops.size(); ++j) {
                const FileOp& o = h->ops[j];
                if (j == i || o.type == OP_STEM) continue;
                const bool reads = o.src_buf == p3.dst_buf && o.src_choff < p3.dst_choff + p3.dst_c && p3.dst_choff < o.src_choff + o.src_c;
                const bool reads_res = o.res_buf == p3.dst_buf && o.res_choff < p3.dst_choff + p3.dst_c && p3.dst_choff < o.res_choff + o.dst_c;
                if (reads || reads_res) { ++n_readers; reader = reads && !reads_res ? (int)j : -2; }
            }
            if (n_readers != 1 || reader <= (int)i) continue;
            const FileOp& p1 = h->ops[reader];
            if (p1.type != OP_CONV || h->convs[p1.conv].k != 1 || h->convs[p1.conv].s != 1 || p1.res_buf >= 0) continue;
            if (h->fuse_up[reader] >= 0 || p1.src_buf != p3.dst_buf) continue;
            const int lead = p3.dst_choff - p1.src_choff;
            const bool exact = lead == 0 && p1.src_c == p3.dst_c;                                   // reads exactly that slice
            const bool tail = fuse_tail && lead > 0 && (lead % 16) == 0 && (p3.dst_c % 16) == 0 && h->convs[p3.conv].s == 1 &&
                              p1.src_choff + p1.src_c == p3.dst_choff + p3.dst_c;                    // its input slice ENDS with that slice
            if (!(exact && p3.res_buf < 0) && !(fuse_tail && (exact || tail))) continue;
            h->fuse2[i] = reader; h->fuse2_lead[i] = exact ? 0 : lead;
        }
    }
    if (!h->host_only) {
        float lut[256];
        for (int i = 0; i < 256; ++i) lut[i] = (float)i / 255.0f;     // im /= 255 (IEEE fp32 division)
        HIPCHK(hipMalloc(&h->lut, sizeof(lut)));
        HIPCHK(hipMemcpy(h->lut, lut, sizeof(lut), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&h->zeros, 256));
        HIPCHK(hipMemset(h->zeros, 0, 256));
    }
    if (const char* e = getenv("MI355_STREAMS")) h->n_streams = std::max(1, std::min(8, atoi(e)));
    if (const char* e = getenv("MI355_STREAMS_MAX_BATCH")) h->streams_max_batch = atoi(e);
    if (const char* e = getenv("MI355_STREAMS_MIN_BATCH")) h->streams_min_batch = atoi(e);
    if (const char* e = getenv("MI355_GROUPS")) h->use_groups = atoi(e);
    if (const char* e = getenv("MI355_MEM_REUSE")) h->mem_reuse = atoi(e);
    if (const char* e = getenv("MI355_GROUP_MAX_BATCH")) h->group_max_batch = atoi(e);
    return build_schedule(h);
}

// dependency DAG of the op program -> launch order + stream assignment (see mi355_yolo::n_streams)
static int build_schedule(mi355_yolo* h) {
    const int n = (int)h->ops.size();
    auto written = [&](const FileOp& o) { return o.type == OP_SPPF_POOL ? 3 * o.src_c : o.dst_c; };
    auto overlaps = [](int a0, int ac, int b0, int bc) { return a0 < b0 + bc && b0 < a0 + ac; };
    std::vector<std::vector<int>> deps(n);
    auto add_readers_deps = [&](int i, int buf, int off, int c) {
        for (int j = 0; j < i; ++j) {
            const FileOp& w = h->ops[j];
            if (w.dst_buf == buf && overlaps(w.dst_choff, written(w), off, c)) deps[i].push_back(j);
        }
    };
    for (int i = 0; i < n; ++i) {
        const FileOp& o = h->ops[i];
        if (o.type != OP_STEM) add_readers_deps(i, o.src_buf, o.src_choff, o.src_c);
        if (o.res_buf >= 0) add_readers_deps(i, o.res_buf, o.res_choff, o.dst_c);
        if (h->fuse_up[i] >= 0) {                   // may read the upsample's source directly (fused) or its output (not fused)
            const FileOp& u = h->ops[h->fuse_up[i]];
            add_readers_deps(i, u.src_buf, u.src_choff, u.src_c);
        }
        if (h->fuse2[i] >= 0 && h->fuse2_lead[i] > 0) {      // its fused pointwise stage reads the lead slices of the concat buffer
            const FileOp& p1 = h->ops[h->fuse2[i]];
            add_readers_deps(i, p1.src_buf, p1.src_choff, h->fuse2_lead[i]);
        }
        std::sort(deps[i].begin(), deps[i].end());
        deps[i].erase(std::unique(deps[i].begin(), deps[i].end()), deps[i].end());
    }
    h->deps = deps;
    {   // transitive RAW ancestors (program order is a topological order: producers precede their readers)
        const size_t words = ((size_t)n + 63) / 64;
        h->anc.assign(n, std::vector<unsigned long long>(words, 0ull));
        for (int i = 0; i < n; ++i)
            for (int d : deps[i]) {
                h->anc[i][d >> 6] |= 1ull << (d & 63);
                for (size_t w = 0; w < words; ++w) h->anc[i][w] |= h->anc[d][w];
            }
    }
    std::vector<int> depth(n, 0);
    for (int i = 0; i < n; ++i)
        for (int d : deps[i]) depth[i] = std::max(depth[i], depth[d] + 1);
    h->sched_order.resize(n);
    for (int i = 0; i < n; ++i) h->sched_order[i] = i;
    std::stable_sort(h->sched_order.begin(), h->sched_order.end(), [&](int a, int b) { return depth[a] < depth[b]; });
    h->op_stream.assign(n, 0);
    std::vector<char> claimed(n, 0);
    int rr = 0;
    for (int idx : h->sched_order) {
        int from = -1;
        for (int k = (int)deps[idx].size() - 1; k >= 0; --k)        // the most recent producer whose stream is still free to continue
            if (!claimed[deps[idx][k]]) { from = deps[idx][k]; break; }
        if (deps[idx].empty()) h->op_stream[idx] = 0;
        else if (from >= 0) { h->op_stream[idx] = h->op_stream[from]; claimed[from] = 1; }
        else h->op_stream[idx] = h->n_streams > 1 ? 1 + (rr++ % (h->n_streams - 1)) : 0;
    }
    h->leaf_ops.clear();
    for (int i = 0; i < n; ++i)
        for (const FileLevel& lv : h->levels)
            if ((int)lv.buf == h->ops[i].dst_buf) { h->leaf_ops.push_back(i); break; }
    h->op_xdeps.assign(n, {});
    h->op_signals.assign(n, 0);
    for (int i = 0; i < n; ++i)
        for (int d : deps[i])
            if (h->op_stream[d] != h->op_stream[i]) { h->op_xdeps[i].push_back(d); h->op_signals[d] = 1; }
    for (int l : h->leaf_ops) if (h->op_stream[l] != 0) h->op_signals[l] = 1;
    h->op_done.assign(n, nullptr);
    if (h->host_only) return MI355_OK;
    for (int i = 0; i < n; ++i)
        if (h->op_signals[i]) HIPCHK(hipEventCreateWithFlags(&h->op_done[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    h->aux.assign(std::max(0, h->n_streams - 1), nullptr);
    for (auto& st : h->aux) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return MI355_OK;
}

// ---- persisted launch-plan choices ---------------------------------------------------------------------------
// Timing 32 candidates x 60-90 convs costs a second or two per (frames, H, W); eight ranks of one node (and every later
// process) need not repeat it.  The autotuner's CHOICES -- an index into each conv's candidate list -- are kept in a small
// text file keyed by (model image hash, precision, frames, H, W, planner version); a file whose candidate counts do not
// match the running planner is ignored.  MI355_PLAN_CACHE=<dir> moves the directory, MI355_PLAN_CACHE=0 turns it off.
static const char* kPlanVersion = "mi355-plans-r03a";

// What a persisted choice (an INDEX into a candidate list) means depends on the lists themselves: the planner build, its env
// knobs, the GPU.  The file therefore carries a fingerprint of every candidate's launch geometry plus the device's arch name
// and CU count; a file written by another build / device / knob setting does not match and is ignored (then overwritten).
static unsigned long long fnv1a(unsigned long long h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static unsigned long long cand_fingerprint(unsigned long long hsh, const std::vector<ConvLaunch>& list) {
    for (const ConvLaunch& l : list) {
        const int f[] = {l.version, l.CT, l.PT, l.WP, l.a.TW, l.a.TH, l.a.ck, l.a.cgroups, (int)l.lds, (int)l.grid_x, (int)l.grid_y, l.a.w2 ? 1 : 0, l.a.up_c};
        hsh = fnv1a(hsh, f, sizeof(f));
    }
    const int end = -1;
    return fnv1a(hsh, &end, sizeof(end));
}

static std::string plan_cache_path(const mi355_yolo* h, int nb, int Hl, int Wl) {
    const char* e = getenv("MI355_PLAN_CACHE");
    if (e && std::strcmp(e, "0") == 0) return std::string();
    std::string dir;
    if (e && *e) dir = e;
    else if (const char* home = getenv("HOME")) dir = std::string(home) + "/.cache/mi355yolo";
    else dir = "/tmp/mi355yolo";
    (void)mkdir(dir.substr(0, dir.rfind('/')).c_str(), 0755);
    (void)mkdir(dir.c_str(), 0755);
    char name[160];
    snprintf(name, sizeof(name), "/%016llx_%s_t%d_%dx%dx%d.plan", h->model_hash, h->half ? "f16" : "f32", h->autotune, nb, Hl, Wl);
    return dir + name;
}

constexpr int kUpBase = 10000;       // chosen[i] >= kUpBase: the conv reads the upsample kernel's output with plan chosen[i] - kUpBase

static bool load_plan_choices(const mi355_yolo* h, int nb, int Hl, int Wl, const std::vector<int>& n_cands, unsigned long long fp,
                              std::vector<int>* chosen, std::vector<int>* gsel) {
    const std::string path = plan_cache_path(h, nb, Hl, Wl);
    if (path.empty()) return false;
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char ver[64] = {0};
    int n = 0;
    unsigned long long got_fp = 0;
    bool ok = std::fscanf(f, "%63s %d %llx", ver, &n, &got_fp) == 3 && std::strcmp(ver, kPlanVersion) == 0 && n == (int)n_cands.size() && got_fp == fp;
    std::vector<int> got(n_cands.size(), 0), gg(n_cands.size(), -1);
    for (size_t i = 0; ok && i < n_cands.size(); ++i) {
        int c = 0, nc = 0, g = -1;
        // nc = plain + 1000 * fused-pointwise + 1000000 * behind-the-upsample-kernel list sizes.  c in [0, 10000): index into the
        // plain list; c < 0: fused plan -(c + 1); c >= 10000: plan c - 10000 of the list that reads the upsample kernel's output
        // third field: the plan this op runs with INSIDE its step's grouped launch (index into the list `c` selects from), or -1
        ok = std::fscanf(f, "%d/%d/%d", &c, &nc, &g) == 3 && nc == n_cands[i] && g >= -1 && g < 1000 &&
             (c >= kUpBase ? (c - kUpBase < nc / 1000000) : c >= 0 ? (c < nc % 1000 || nc == 0) : (-c - 1 < (nc / 1000) % 1000));
        got[i] = c; gg[i] = g;
    }
    std::fclose(f);
    if (ok) { *chosen = got; *gsel = gg; }
    return ok;
}

static void save_plan_choices(const mi355_yolo* h, int nb, int Hl, int Wl, const std::vector<int>& n_cands, unsigned long long fp,
                              const std::vector<int>& chosen, const std::vector<int>& gsel) {
    const std::string path = plan_cache_path(h, nb, Hl, Wl);
    if (path.empty()) return;
    const std::string tmp = path + "." + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "w");
    if (!f) return;
    std::fprintf(f, "%s %zu %llx\n", kPlanVersion, n_cands.size(), fp);
    for (size_t i = 0; i < n_cands.size(); ++i) std::fprintf(f, "%d/%d/%d\n", chosen[i], n_cands[i], gsel[i]);
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());   // atomic: ranks may race
}

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

static int launch_net(mi355_yolo* h, Prof& pf, const uint8_t* stem_in, int nb, const Geometry& g, bool full_pred);

// Liveness-based placement of the activation buffers in ONE arena (host arithmetic only).  Fills h->dbuf_cs / dbuf_es.
static void plan_memory(mi355_yolo* h, int nb, int Hl, int Wl, std::vector<size_t>* off_out, std::vector<size_t>* bytes_out,
                        size_t* arena_out, size_t* plain_out) {
    const size_t nbufs = h->bufs.size();
    h->dbuf_cs.assign(nbufs, 0); h->dbuf_es.assign(nbufs, 4);
    std::vector<size_t> bytes(nbufs, 0);
    std::vector<char> pinned(nbufs, 0);
    for (size_t i = 0; i < nbufs; ++i) {
        // half=True: every buffer holds fp16 except the head outputs (raw box / class / keypoint logits), which the
        // final 1x1 convs write in fp32 for the decode kernel
        bool is_head = false;
        for (const FileLevel& lv : h->levels) is_head |= (lv.buf == i);
        const int es = (h->half && !is_head) ? 2 : 4;
        const int cs = round_up((int)h->bufs[i].channels, 16 / es);
        h->dbuf_es[i] = es; h->dbuf_cs[i] = cs;
        static const size_t arena_align = getenv("MI355_ARENA_ALIGN") ? (size_t)std::max(256, atoi(getenv("MI355_ARENA_ALIGN"))) : 256;
        bytes[i] = round_up_sz((size_t)nb * (Hl / h->bufs[i].stride_div) * (Wl / h->bufs[i].stride_div) * cs * es, arena_align);
        // bytes of its own, forever: head outputs (the decode kernel reads them after the last op) and buffers with pad
        // channels (cs > channels: zeroed once here, read -- times zero weights -- by the convs' padded k-blocks, never
        // written: another tensor's bits there could be NaN patterns)
        pinned[i] = is_head || cs != (int)h->bufs[i].channels || !h->mem_reuse;
    }
    // ---- liveness-based placement.  A buffer's users = every op that reads or writes any of its channels (a conv that may
    // read an upsample's SOURCE directly counts as a user of that source).  Buffer b may take bytes of buffer a iff EVERY
    // user of a is a RAW ancestor of EVERY writer of b: then the write-after-read / write-after-write order between them is
    // implied by the dependencies the schedulers already honour (streams along the DAG, steps, groups) -- no edge is added,
    // no parallelism is lost, and the sharing is race-free under any schedule that respects RAW.
    const int n_ops = (int)h->ops.size();
    std::vector<std::vector<int>> users(nbufs), writers(nbufs);
    for (int i = 0; i < n_ops; ++i) {
        const FileOp& o = h->ops[i];
        auto use = [&](int b) { if (b >= 0 && (users[b].empty() || users[b].back() != i)) users[b].push_back(i); };
        if (o.type != OP_STEM) use(o.src_buf);
        if (o.res_buf >= 0) use(o.res_buf);
        use(o.dst_buf); writers[o.dst_buf].push_back(i);
        if (h->fuse_up[i] >= 0) use(h->ops[h->fuse_up[i]].src_buf);
        // a pointwise conv that may run INSIDE this op's launch (Conv3x3 -> Conv1x1 fused): its output is then written while
        // this op still reads its own inputs, so this op counts as a writer (and user) of that output buffer as well
        if (h->fuse2[i] >= 0) { const int d2 = h->ops[h->fuse2[i]].dst_buf; use(d2); writers[d2].push_back(i); }
    }
    auto is_anc = [&](int a, int of) { return (h->anc[of][a >> 6] >> (a & 63)) & 1ull; };
    auto may_share = [&](size_t a, size_t b) {          // may b (written later) take a's bytes?
        if (pinned[a] || pinned[b] || writers[b].empty() || users[a].empty()) return false;
        for (int w : writers[b])
            for (int u : users[a]) if (!is_anc(u, w)) return false;
        return true;
    };
    std::vector<size_t> order(nbufs), off(nbufs, 0);
    for (size_t i = 0; i < nbufs; ++i) order[i] = i;
    auto first_w = [&](size_t b) { return writers[b].empty() ? 1 << 30 : writers[b].front(); };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return first_w(a) < first_w(b); });
    size_t arena_bytes = 0, plain_bytes = 0;
    std::vector<size_t> placed;
    for (size_t b : order) {
        plain_bytes += bytes[b];
        // lowest offset where b overlaps only buffers whose bytes it may take (first fit over the sorted conflict list)
        std::vector<std::pair<size_t, size_t>> busy;
        for (size_t a : placed) if (!may_share(a, b)) busy.push_back({off[a], off[a] + bytes[a]});
        std::sort(busy.begin(), busy.end());
        size_t at = 0;
        for (const auto& iv : busy) {
            if (at + bytes[b] <= iv.first) break;
            at = std::max(at, iv.second);
        }
        off[b] = at; placed.push_back(b);
        arena_bytes = std::max(arena_bytes, at + bytes[b]);
    }
    *off_out = off;
    if (bytes_out) *bytes_out = bytes;
    *arena_out = arena_bytes; *plain_out = plain_bytes;
}

static int ensure_shape(mi355_yolo* h, int nb, int Hl, int Wl) {
    if (h->cur_nb == nb && h->cur_H == Hl && h->cur_W == Wl) return MI355_OK;
    if ((Hl % 32) || (Wl % 32)) return fail(MI355_EINVAL, "letterboxed size must be a multiple of 32");
    const size_t nbufs = h->bufs.size();
    if (h->cur_H != Hl || h->cur_W != Wl || nb > h->alloc_nb) {
        // (re)allocate: a new letterboxed size, or more frames per pass than the buffers hold
        h->free_shape();
        h->dbuf.assign(nbufs, nullptr);
        std::vector<size_t> off;
        size_t arena_bytes = 0, plain_bytes = 0;
        plan_memory(h, nb, Hl, Wl, &off, nullptr, &arena_bytes, &plain_bytes);
        HIPCHK(hipMalloc(&h->arena, std::max<size_t>(arena_bytes, 256)));
        HIPCHK(hipMemsetAsync(h->arena, 0, arena_bytes, h->stream));       // pad channels (pinned buffers) stay zero forever
        for (size_t i = 0; i < nbufs; ++i) h->dbuf[i] = (float*)(h->arena + off[i]);
        h->act_bytes = (long long)arena_bytes; h->act_bytes_noreuse = (long long)plain_bytes;
        int A = 0;
        for (const FileLevel& lv : h->levels) A += (Hl / lv.stride) * (Wl / lv.stride);
        h->A = A; h->Apow2 = 1; while (h->Apow2 < A) h->Apow2 <<= 1;
        HIPCHK(hipMalloc(&h->pred, (size_t)nb * A * h->no() * 4));
        HIPCHK(hipMalloc(&h->best, (size_t)nb * A * sizeof(float2)));
        HIPCHK(hipMalloc(&h->keys, (size_t)nb * h->Apow2 * 8));
        HIPCHK(hipMalloc(&h->lbox, (size_t)nb * Hl * Wl * 3));
        h->alloc_nb = nb; h->cur_H = Hl; h->cur_W = Wl;
    }
    // ---- launch plans for nb frames per pass on the existing buffers ----
    h->cur_nb = 0;                      // a rebuild that fails half-way is retried by the next call instead of running stale plans
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.second);      // captured launches embed the old plans
    h->graphs.clear();
    h->plans.assign(h->ops.size(), ConvLaunch{});
    const bool tune_log = getenv("MI355_TUNE_LOG") != nullptr;
    const std::array<int, 3> shape_key{nb, Hl, Wl};
    // pass 1: the candidate lists (host work only).  A 3x3 conv that may absorb its pointwise consumer (fuse2) gets two lists:
    // the plain one and the fused one; which form runs is decided below, by the stopwatch.
    std::fill(h->skip_op.begin(), h->skip_op.end(), 0);
    std::vector<std::vector<ConvLaunch>> cands(h->ops.size()), cands_f(h->ops.size()), cands_u(h->ops.size());
    std::vector<int> n_cands(h->ops.size(), 0);
    const size_t top = (size_t)std::max(1, h->autotune);
    for (size_t i = 0; i < h->ops.size(); ++i) {
        const FileOp& o = h->ops[i];
        if (o.type != OP_CONV) continue;
        const FileConv& c = h->convs[o.conv];
        ConvArgs a{};
        const int sd_in = h->bufs[o.src_buf].stride_div, sd_out = h->bufs[o.dst_buf].stride_div;
        a.src = h->view(o.src_buf, o.src_choff); a.src_cs = h->dbuf_cs[o.src_buf];
        a.dst = h->view(o.dst_buf, o.dst_choff); a.dst_cs = h->dbuf_cs[o.dst_buf];
        if (o.res_buf >= 0) { a.res = h->view(o.res_buf, o.res_choff); a.res_cs = h->dbuf_cs[o.res_buf]; }
        if (h->half) {
            a.dtype = 1; a.out_f32 = h->dbuf_es[o.dst_buf] == 4;
            if (h->dbuf_es[o.src_buf] != 2 || (o.res_buf >= 0 && h->dbuf_es[o.res_buf] != h->dbuf_es[o.dst_buf]))
                return fail(MI355_EFORMAT, "half: a conv reads a head output buffer");
        }
        a.wpk = h->dconv[o.conv].wpk; a.bias = h->dconv[o.conv].bias; a.zeros = h->zeros;
        a.B = nb; a.Hin = Hl / sd_in; a.Win = Wl / sd_in; a.Hout = Hl / sd_out; a.Wout = Wl / sd_out;
        a.Cin = c.cin; a.Cout = c.cout; a.k = c.k; a.stride = c.s; a.pad = c.pad; a.act = c.act;
        if (a.Hout * (int)c.s != a.Hin || a.Wout * (int)c.s != a.Win) return fail(MI355_EFORMAT, "conv resolution mismatch in program");
        if (h->fuse_up[i] >= 0) {
            const FileOp& u = h->ops[h->fuse_up[i]];
            ConvArgs f = a;
            f.src2 = h->view(u.src_buf, u.src_choff); f.src2_cs = h->dbuf_cs[u.src_buf]; f.up_c = u.src_c;
            const bool same_prec = h->dbuf_es[u.src_buf] == h->dbuf_es[o.src_buf];
            const bool shape_ok = h->bufs[u.src_buf].stride_div == 2 * sd_in && (a.Hin % 2) == 0 && (a.Win % 2) == 0 &&
                                  (u.src_c % (16 / h->dbuf_es[o.src_buf])) == 0;
            if (same_prec && shape_ok && plan_conv_candidates(f, &cands[i]) == nullptr && !cands[i].empty()) {
                // the other form -- upsample kernel, then any pointwise plan on its output -- competes on the stopwatch below
                static const bool up_tune = !getenv("MI355_UPSAMPLE_TUNE") || atoi(getenv("MI355_UPSAMPLE_TUNE")) != 0;
                if (!up_tune || plan_conv_candidates(a, &cands_u[i]) != nullptr) cands_u[i].clear();
                if (cands_u[i].size() > top) cands_u[i].resize(top);
                a = f;
                h->fused_away[h->fuse_up[i]] = 1;
            } else {                                    // no v4 plan for this shape: run the upsample kernel after all
                cands[i].clear();
                h->fused_away[h->fuse_up[i]] = 0;
            }
        }
        if (h->fuse2[i] >= 0) {
            const FileOp& o1 = h->ops[h->fuse2[i]];
            const FileConv& c1 = h->convs[o1.conv];
            ConvArgs f = a;
            f.f2_wpk = h->dconv[o1.conv].wpk; f.f2_bias = h->dconv[o1.conv].bias;
            f.f2_dst = h->view(o1.dst_buf, o1.dst_choff); f.f2_dst_cs = h->dbuf_cs[o1.dst_buf]; f.f2_cout = c1.cout; f.f2_act = c1.act;
            f.f2_out_f32 = (h->half && h->dbuf_es[o1.dst_buf] == 4) ? 1 : 0;
            if (h->fuse2_lead[i] > 0) { f.f2_lead = h->view(o1.src_buf, o1.src_choff); f.f2_lead_cs = h->dbuf_cs[o1.src_buf]; f.f2_lead_c = h->fuse2_lead[i]; }
            if (plan_conv_candidates(f, &cands_f[i]) != nullptr) cands_f[i].clear();
            if (cands_f[i].size() > top) cands_f[i].resize(top);
        }
        if (cands[i].empty()) KCHK(plan_conv_candidates(a, &cands[i]));
        if (cands[i].size() > top) cands[i].resize(top);
        n_cands[i] = (int)cands[i].size() + 1000 * (int)cands_f[i].size() + 1000000 * (int)cands_u[i].size();
        h->plans[i] = cands[i][0];
    }
    // pass 2: choices -- this process's memory, then the plan file, then the stopwatch.  chosen[i] >= 0: index into the plain
    // list; chosen[i] = -(k + 1): fused plan k (the pointwise consumer then runs inside this launch and is skipped).
    const size_t n_ops = h->ops.size();
    std::vector<int> chosen(n_ops, 0), gsel(n_ops, -1);
    bool have = false, have_groups = false;       // have_groups: gsel is a decision (memory / file), not the initial "none"
    for (const auto& t : h->tuned)
        if (t.first == shape_key) {
            chosen.assign(t.second.begin(), t.second.begin() + n_ops); gsel.assign(t.second.begin() + n_ops, t.second.end());
            have = have_groups = true;
        }
    unsigned long long fp = 1469598103934665603ull;
    {
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) {
            fp = fnv1a(fp, prop.gcnArchName, std::strlen(prop.gcnArchName));
            fp = fnv1a(fp, &prop.multiProcessorCount, sizeof(int));
        }
        for (size_t i = 0; i < h->ops.size(); ++i) { fp = cand_fingerprint(fp, cands[i]); fp = cand_fingerprint(fp, cands_f[i]); fp = cand_fingerprint(fp, cands_u[i]); }
        // the scheduling regime the choices were made for (a file written with grouped launches off must not pin "no groups")
        const int regime[3] = {h->use_groups, h->group_max_batch, h->streams_min_batch};
        fp = fnv1a(fp, regime, sizeof(regime));
    }
    h->plan_source = have ? 1 : 0;          // 1 = this process's memory
    if (!have && h->autotune) { have = load_plan_choices(h, nb, Hl, Wl, n_cands, fp, &chosen, &gsel); if (have) { h->plan_source = 2; have_groups = true; } }
    auto run_upsample = [&](int ui) -> int {
        const FileOp& u = h->ops[ui];
        const int sd_in = h->bufs[u.src_buf].stride_div, dv = h->dbuf_es[u.src_buf] == 2 ? 2 : 1;
        KCHK(launch_upsample2x(h->view(u.src_buf, u.src_choff), h->dbuf_cs[u.src_buf] / dv, h->view(u.dst_buf, u.dst_choff),
                               h->dbuf_cs[u.dst_buf] / dv, nb, Hl / sd_in, Wl / sd_in, u.src_c / dv, h->stream));
        return MI355_OK;
    };
    // Short launches (a few frames per pass) are timed IN CONTEXT: a train of 8 x [spacer, candidate], where the spacer is the
    // launch that precedes the candidate in the net (its producer, as a rule).  A train of one kernel alone flatters it -- its
    // input lines, its code and its weights are hot in the caches of the CUs that just ran the same thing -- and flatters
    // persistent / prefetching kernels most: timed that way the autotuner picked the pipelined pointwise kernel (19 us in the
    // net, 10 in its train) over the streaming one; with MI355_CONV_V4=0 batch 1 ran 7 % faster.  The spacer's own train time
    // is subtracted for the log; decisions between alternatives (fused or not, grouped or not) time whole sequences.
    const ConvLaunch* spacer = nullptr;
    float spacer_ms = -1.f;
    auto time_train = [&](const std::function<int()>& body, float* ms_out) -> int {
        float ms = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            HIPCHK(hipEventRecord(h->ev0, h->stream));
            for (int j = 0; j < 8; ++j) { const int rc = body(); if (rc) return rc; }
            HIPCHK(hipEventRecord(h->ev1, h->stream));
            HIPCHK(hipEventSynchronize(h->ev1));
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, h->ev0, h->ev1));
            ms = std::min(ms, t / 8.0f);
        }
        *ms_out = ms;
        return MI355_OK;
    };
    auto launch = [&](const ConvLaunch& l) -> int { KCHK(run_conv(l, h->stream)); return MI355_OK; };
    auto time_list = [&](const std::vector<ConvLaunch>& list, const char* name, int* best_k, float* best_ms) -> int {
        // Time launch plans on the real buffers (outputs are overwritten by the next real pass; the accumulation order is
        // plan-independent, so the choice cannot change results).
        *best_ms = 1e30f; *best_k = 0;
        for (size_t k = 0; k < list.size(); ++k) {
            float ms = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                HIPCHK(hipEventRecord(h->ev0, h->stream));
                KCHK(run_conv(list[k], h->stream));
                HIPCHK(hipEventRecord(h->ev1, h->stream));
                HIPCHK(hipEventSynchronize(h->ev1));
                float t = 0.f;
                HIPCHK(hipEventElapsedTime(&t, h->ev0, h->ev1));
                if (rep > 0) ms = std::min(ms, t);          // first run warms the instruction cache
            }
            if (ms < 0.1f) {
                // short launches (small batches): a single 5-20 us launch is at the resolution of the event pair, and candidates
                // differ by fractions of a microsecond -- time trains of 8 back-to-back launches (as they run in the net) instead
                if (spacer && spacer_ms < 0.f) { const int rc = time_train([&] { return launch(*spacer); }, &spacer_ms); if (rc) return rc; }
                const int rc = time_train([&] { if (spacer) { const int r = launch(*spacer); if (r) return r; } return launch(list[k]); }, &ms);
                if (rc) return rc;
                if (spacer) ms = std::max(ms - spacer_ms, 1e-4f);
            }
            if (ms < *best_ms) { *best_ms = ms; *best_k = (int)k; }
            if (tune_log)
                fprintf(stderr, "[tune] %-40s v%d CT%d PT%d WP%d G%d%s tile %dx%d ck%d lds %zu grid %ux%u : %.1f us  %.1f TFLOP/s\n",
                        name, list[k].version, list[k].CT, list[k].PT, list[k].WP, list[k].a.cgroups, list[k].a.w2 ? " +1x1" : "", list[k].a.TW,
                        list[k].a.TH, list[k].a.ck, list[k].lds, list[k].grid_x, list[k].grid_y, ms * 1e3, list[k].flops / (ms * 1e-3) / 1e12);
        }
        return MI355_OK;
    };
    std::vector<ConvLaunch> finals(h->ops.size());
    // both forms of every binary decision (index into `chosen`'s encoding; -9999 = the form does not exist), for the pass-level check below
    constexpr int kNone = -9999;
    std::vector<int> alt_fused(h->ops.size(), kNone), alt_sep(h->ops.size(), kNone), alt_read(h->ops.size(), kNone), alt_up(h->ops.size(), kNone);
    const std::vector<char> fused_away_base(h->fused_away.begin(), h->fused_away.end());     // as pass 1 left it (upsample read through the conv where possible)
    if (!have && h->autotune) {
        std::vector<char> done(h->ops.size(), 0);
        for (size_t i = 0; i < h->ops.size(); ++i) {
            if (h->ops[i].type != OP_CONV || done[i]) continue;
            const char* name = h->convs[h->ops[i].conv].name;
            int k = 0; float ms = 0.f;
            if (cands[i].size() > 1 || !cands_f[i].empty()) { const int rc = time_list(cands[i], name, &k, &ms); if (rc) return rc; }
            chosen[i] = k;
            if (!cands_u[i].empty()) {
                // upsample fused into the read side vs upsample kernel + best plan on its output
                const int ui = h->fuse_up[i];
                int ku = 0; float msu = 0.f, msk = 1e30f;
                const int rc = time_list(cands_u[i], name, &ku, &msu); if (rc) return rc;
                for (int rep = 0; rep < 3; ++rep) {
                    HIPCHK(hipEventRecord(h->ev0, h->stream));
                    const int rcu = run_upsample(ui); if (rcu) return rcu;
                    HIPCHK(hipEventRecord(h->ev1, h->stream));
                    HIPCHK(hipEventSynchronize(h->ev1));
                    float t = 0.f;
                    HIPCHK(hipEventElapsedTime(&t, h->ev0, h->ev1));
                    if (rep > 0) msk = std::min(msk, t);
                }
                bool separate = msu + msk < ms;
                if (ms < 0.1f) {                  // short launches: time both sequences as they would run
                    float ta = 0.f, tb = 0.f;
                    int r2 = time_train([&] { if (spacer) { const int r = launch(*spacer); if (r) return r; } return launch(cands[i][k]); }, &ta); if (r2) return r2;
                    r2 = time_train([&] { if (spacer) { const int r = launch(*spacer); if (r) return r; } const int r = run_upsample(ui); if (r) return r; return launch(cands_u[i][ku]); }, &tb); if (r2) return r2;
                    separate = tb < ta;
                }
                alt_read[i] = k; alt_up[i] = kUpBase + ku;
                if (separate) chosen[i] = kUpBase + ku;
                if (tune_log) fprintf(stderr, "[tune] %s: upsample on read %.1f us vs upsample kernel %.1f + conv %.1f us -> %s\n", name, ms * 1e3,
                                      msk * 1e3, msu * 1e3, chosen[i] >= kUpBase ? "separate" : "fused");
            }
            if (!cands_f[i].empty()) {
                // fused vs separate: best fused launch against best 3x3 + best pointwise launch
                const int j = h->fuse2[i];
                int kf = 0, kj = 0; float msf = 0.f, msj = 0.f;
                int rc = time_list(cands_f[i], name, &kf, &msf); if (rc) return rc;
                const ConvLaunch* outer = spacer; const float outer_ms = spacer_ms;
                const ConvLaunch first = chosen[i] >= kUpBase ? cands_u[i][chosen[i] - kUpBase] : cands[i][chosen[i]];
                spacer = &first; spacer_ms = -1.f;                       // the pointwise conv's producer is this conv
                rc = time_list(cands[j], h->convs[h->ops[j].conv].name, &kj, &msj);
                spacer = outer; spacer_ms = outer_ms;
                if (rc) return rc;
                chosen[j] = kj; done[j] = 1;
                bool fuse = msf < ms + msj;
                if (ms < 0.1f && chosen[i] < kUpBase) {                   // short launches: time both sequences as they would run
                    float ta = 0.f, tb = 0.f;
                    rc = time_train([&] { if (spacer) { const int r = launch(*spacer); if (r) return r; } return launch(cands_f[i][kf]); }, &ta); if (rc) return rc;
                    rc = time_train([&] { if (spacer) { const int r = launch(*spacer); if (r) return r; } const int r = launch(first); if (r) return r; return launch(cands[j][kj]); }, &tb); if (rc) return rc;
                    fuse = ta < tb;
                }
                alt_sep[i] = chosen[i]; alt_fused[i] = -(kf + 1);
                if (fuse) chosen[i] = -(kf + 1);
                if (tune_log) fprintf(stderr, "[tune] %s: fused %.1f us vs separate %.1f + %.1f us -> %s\n", name, msf * 1e3, ms * 1e3, msj * 1e3,
                                      chosen[i] < 0 ? "fused" : "separate");
            }
            // the launch that will precede the next op in the net becomes the next spacer
            finals[i] = chosen[i] < 0 ? cands_f[i][-chosen[i] - 1] : chosen[i] >= kUpBase ? cands_u[i][chosen[i] - kUpBase] : cands[i][chosen[i]];
            spacer = &finals[i];
            if (h->fuse2[i] >= 0 && chosen[i] >= 0 && done[h->fuse2[i]]) {
                const int j = h->fuse2[i];
                finals[j] = cands[j][chosen[j]];
                spacer = &finals[j];
            }
            spacer_ms = -1.f;
        }
        have = true;
        h->plan_source = 3;
    }
    auto apply_plans = [&]() {                   // `chosen` -> the launch plan of every conv, which ops run inside another's launch
        std::fill(h->skip_op.begin(), h->skip_op.end(), 0);
        std::copy(fused_away_base.begin(), fused_away_base.end(), h->fused_away.begin());
        for (size_t i = 0; i < h->ops.size(); ++i) {
            if (h->ops[i].type != OP_CONV) continue;
            if (chosen[i] < 0 && (size_t)(-chosen[i] - 1) < cands_f[i].size()) {
                h->plans[i] = cands_f[i][-chosen[i] - 1];
                h->skip_op[h->fuse2[i]] = 1;
            } else if (chosen[i] >= kUpBase && (size_t)(chosen[i] - kUpBase) < cands_u[i].size()) {
                h->plans[i] = cands_u[i][chosen[i] - kUpBase];
                h->fused_away[h->fuse_up[i]] = 0;       // the upsample kernel runs; this conv reads its output
            } else if (chosen[i] >= 0 && (size_t)chosen[i] < cands[i].size()) {
                h->plans[i] = cands[i][chosen[i]];
            } else {
                h->plans[i] = cands[i][0];
            }
        }
    };
    if (have) apply_plans();
    // ---- grouped launches of the single-stream regime: list-schedule the launched ops into steps, then decide per step ----
    h->groups.clear(); h->steps.clear();
    h->group_sel.assign(n_ops, -1);
    const bool stepwise = h->use_groups && !h->half && nb <= h->group_max_batch && nb < h->streams_min_batch;
    auto launched = [&](int i) { return !(h->ops[i].type == OP_UPSAMPLE && h->fused_away[i]) && !h->skip_op[i]; };
    auto list_of = [&](int i) -> const std::vector<ConvLaunch>& {      // the candidate list op i's current plan was taken from
        return (h->fuse2[i] >= 0 && h->skip_op[h->fuse2[i]]) ? cands_f[i] : chosen[i] >= kUpBase ? cands_u[i] : cands[i];
    };
    auto build_steps = [&]() -> int {
        h->steps.clear(); h->groups.clear();
        std::fill(h->group_sel.begin(), h->group_sel.end(), -1);
        // avail[i] = step after which op i's output exists; an op is ready when all its producers are available
        const int n = (int)n_ops;
        std::vector<int> avail(n, -1), producer_of(n, -1);
        for (int i = 0; i < n; ++i) if (h->fuse2[i] >= 0 && h->skip_op[h->fuse2[i]]) producer_of[h->fuse2[i]] = i;
        std::vector<char> placed(n, 0);
        int left = 0;
        for (int i = 0; i < n; ++i) left += launched(i);
        // data of an op that is not launched itself: inside its producer's launch (fused pointwise conv) or never materialised
        // (upsample read by its consumer straight from the half-size map: available when the upsample's own producers are)
        std::function<int(int)> avail_of = [&](int d) -> int {
            if (launched(d)) return placed[d] ? avail[d] : 1 << 30;
            if (producer_of[d] >= 0) return placed[producer_of[d]] ? avail[producer_of[d]] : 1 << 30;
            int a = -1;
            for (int dd : h->deps[d]) a = std::max(a, avail_of(dd));
            return a;
        };
        for (int step = 0; left > 0; ++step) {
            mi355_yolo::Step st;
            std::vector<int> now;
            for (int i = 0; i < n; ++i) {
                if (!launched(i) || placed[i]) continue;
                int a = -1;
                for (int d : h->deps[i]) a = std::max(a, avail_of(d));
                if (a < step) now.push_back(i);
            }
            if (now.empty()) return fail(MI355_EFORMAT, "op program has a dependency cycle");
            for (int i : now) { placed[i] = 1; avail[i] = step; --left; st.singles.push_back(i); }
            h->steps.push_back(st);
        }
        return MI355_OK;
    };
    auto tune_groups = [&]() -> int {
        {
            // decide per step by the stopwatch: the convs whose kernel is on the group kernel's menu, as one grid, against
            // the same convs launched one after the other with their individually best plans
            int prev_conv = -1, prev_next = -1;         // a conv launched in the previous step (the spacer of this step's timings)
            for (auto& st : h->steps) {
                prev_conv = prev_next;
                for (int i : st.singles) if (h->ops[i].type == OP_CONV) { prev_next = i; break; }
                struct Member { int op, sel; ConvLaunch l; int kind; float t_ind; };
                std::vector<Member> mem;
                for (int i : st.singles) {
                    if (h->ops[i].type != OP_CONV) continue;
                    const FileConv& c = h->convs[h->ops[i].conv];
                    const std::vector<ConvLaunch>& list = list_of(i);
                    Member m{i, -1, h->plans[i], group_kind(h->plans[i], (int)c.k, (int)c.s), 0.f};
                    int k0 = 0;
                    std::vector<ConvLaunch> one{h->plans[i]};
                    int rc = time_list(one, c.name, &k0, &m.t_ind); if (rc) return rc;
                    if (m.kind >= 0) {
                        for (size_t k = 0; k < list.size(); ++k)
                            if (list[k].fn == m.l.fn && list[k].grid_x == m.l.grid_x && list[k].grid_y == m.l.grid_y && list[k].lds == m.l.lds &&
                                list[k].a.TW == m.l.a.TW && list[k].a.ck == m.l.a.ck && list[k].a.cgroups == m.l.a.cgroups) { m.sel = (int)k; break; }
                    }
                    if (m.sel < 0) {                 // the tuned kernel is not on the menu: the best plan that is
                        std::vector<ConvLaunch> menu; std::vector<int> idx;
                        for (size_t k = 0; k < list.size(); ++k)
                            if (group_kind(list[k], (int)c.k, (int)c.s) >= 0) { menu.push_back(list[k]); idx.push_back((int)k); }
                        if (menu.empty()) continue;
                        int kb = 0; float tb = 0.f;
                        rc = time_list(menu, c.name, &kb, &tb); if (rc) return rc;
                        m.sel = idx[kb]; m.l = menu[kb]; m.kind = group_kind(m.l, (int)c.k, (int)c.s);
                    }
                    mem.push_back(m);
                }
                if (mem.size() < 2) continue;
                std::sort(mem.begin(), mem.end(), [](const Member& a, const Member& b) { return a.t_ind > b.t_ind; });
                if (mem.size() > (size_t)kGroupMax) mem.resize(kGroupMax);
                std::vector<ConvLaunch> ls; std::vector<int> kinds; float t_sum = 0.f;
                for (const Member& m : mem) { ls.push_back(m.l); kinds.push_back(m.kind); t_sum += m.t_ind; }
                GroupLaunch g{};
                if (plan_group(ls, kinds, &g) != nullptr) continue;
                // both forms in context: [the previous step's conv, grouped launch] against [the same conv, the members one by one]
                const ConvLaunch* gsp = prev_conv >= 0 ? &h->plans[prev_conv] : nullptr;
                float t_grp = 1e30f;
                int rcg = time_train([&] { if (gsp) { const int r = launch(*gsp); if (r) return r; } KCHK(run_group(g, h->stream)); return (int)MI355_OK; }, &t_grp);
                if (rcg) return rcg;
                rcg = time_train([&] { if (gsp) { const int r = launch(*gsp); if (r) return r; }
                                       for (const Member& m : mem) { const int r = launch(h->plans[m.op]); if (r) return r; } return (int)MI355_OK; }, &t_sum);
                if (rcg) return rcg;
                static int dbg_seq = 0;
                const char* only = getenv("MI355_GROUP_ONLY");        // debugging: accept only the n-th candidate group
                const bool dbg_ok = !only || atoi(only) == dbg_seq;
                ++dbg_seq;
                if (tune_log) {
                    fprintf(stderr, "[tune] group of %zu:", mem.size());
                    for (const Member& m : mem) fprintf(stderr, " %s(%.1f us; kind %d v%d PT%d CT%d WP%d G%d grid %ux%u lds %zu%s%s)", h->convs[h->ops[m.op].conv].name, m.t_ind * 1e3,
                                                        m.kind, m.l.version, m.l.PT, m.l.CT, m.l.WP, m.l.a.cgroups, m.l.grid_x, m.l.grid_y, m.l.lds, m.l.a.w2 ? " +1x1" : "", m.l.a.res ? " +res" : "");
                    fprintf(stderr, " : grouped %.1f us vs separate %.1f us -> %s\n", t_grp * 1e3, t_sum * 1e3, t_grp < 0.97f * t_sum ? "grouped" : "separate");
                }
                if (t_grp < 0.97f * t_sum && h->use_groups != 2 && dbg_ok)         // MI355_GROUPS=2: step order without grouped launches (debugging)
                    for (const Member& m : mem) gsel[m.op] = m.sel;
            }
        }
        return MI355_OK;
    };
    auto materialise_groups = [&]() {
        // members with a selection leave the step's single launches and form its group
        for (auto& st : h->steps) {
            std::vector<int> members, singles;
            for (int i : st.singles) {
                const bool ok = h->ops[i].type == OP_CONV && gsel[i] >= 0 && (size_t)gsel[i] < list_of(i).size() &&
                                group_kind(list_of(i)[gsel[i]], (int)h->convs[h->ops[i].conv].k, (int)h->convs[h->ops[i].conv].s) >= 0;
                (ok && members.size() < (size_t)kGroupMax ? members : singles).push_back(i);
            }
            if (members.size() < 2) continue;
            std::vector<ConvLaunch> ls; std::vector<int> kinds;
            for (int i : members) {
                const FileConv& c = h->convs[h->ops[i].conv];
                ls.push_back(list_of(i)[gsel[i]]); kinds.push_back(group_kind(ls.back(), (int)c.k, (int)c.s));
            }
            GroupLaunch g{};
            if (plan_group(ls, kinds, &g) != nullptr) continue;
            for (size_t m = 0; m < members.size(); ++m) { g.op[m] = members[m]; h->group_sel[members[m]] = gsel[members[m]]; }
            st.singles = singles; st.group = (int)h->groups.size();
            h->groups.push_back(g);
        }
    };
    // One whole pass of the net (stem .. decode) with the current decisions, in the order and with the launches the product runs:
    // the yardstick for decisions whose effect depends on what runs before and after (a fused launch, a grouped launch).
    auto time_pass = [&](float* ms_out) -> int {
        const Geometry g = make_geometry(Hl, Wl, std::max(Hl, Wl));
        Prof pf{h};
        const bool was = h->profiling; h->profiling = false;
        const int cur = h->cur_nb; h->cur_nb = nb;
        float best = 1e30f;
        int rc = MI355_OK;
        for (int rep = 0; rep < 4 && !rc; ++rep) {
            if (hipEventRecord(h->ev0, h->stream) != hipSuccess) { rc = fail(MI355_EHIP, "event"); break; }
            for (int j = 0; j < 24 && !rc; ++j) rc = launch_net(h, pf, h->lbox, nb, g, false);
            if (rc) break;
            if (hipEventRecord(h->ev1, h->stream) != hipSuccess || hipEventSynchronize(h->ev1) != hipSuccess) { rc = fail(MI355_EHIP, "event"); break; }
            float t = 0.f;
            (void)hipEventElapsedTime(&t, h->ev0, h->ev1);
            if (rep > 0) best = std::min(best, t / 24.0f);
        }
        h->profiling = was; h->cur_nb = cur;
        *ms_out = best;
        return rc;
    };
    if (stepwise) {
        int rc = build_steps(); if (rc) return rc;
        const bool fresh = !have_groups && have && h->autotune;
        static const int pass_tune = getenv("MI355_PASS_TUNE") ? atoi(getenv("MI355_PASS_TUNE")) : 1;
        if (fresh && h->plan_source == 3 && pass_tune) {
            // Pass-level check of the binary decisions (fp32, latency-bound regime).  The per-op stopwatch compares a fused launch
            // with its two halves in a train of their own; what the choice does to the PASS -- caches, the launch behind it, the
            // steps it merges or splits -- shows only there.  Greedy: flip one decision, time whole passes, keep what is faster.
            float best = 0.f;
            rc = time_pass(&best); if (rc) return rc;
            auto try_flip = [&](size_t i, int a, int b, const char* what) -> int {
                if (a == kNone || b == kNone) return MI355_OK;
                const int old = chosen[i], alt = old == a ? b : a;
                chosen[i] = alt;
                apply_plans();
                int r = build_steps(); if (r) return r;
                float t = 0.f;
                r = time_pass(&t); if (r) return r;
                const bool keep = t < 0.997f * best;
                if (tune_log) fprintf(stderr, "[tune] pass check %s %s: %.1f us -> %.1f us per pass: %s\n", h->convs[h->ops[i].conv].name, what, best * 1e3, t * 1e3,
                                      keep ? "flipped" : "kept");
                if (keep) best = t; else { chosen[i] = old; apply_plans(); r = build_steps(); if (r) return r; }
                return MI355_OK;
            };
            for (size_t i = 0; i < n_ops; ++i) {
                rc = try_flip(i, alt_fused[i], alt_sep[i], "fused <-> separate"); if (rc) return rc;
                if (chosen[i] >= 0) { rc = try_flip(i, alt_read[i], alt_up[i], "upsample on read <-> kernel"); if (rc) return rc; }
            }
        }
        if (fresh) {
            rc = tune_groups(); if (rc) return rc;
            have_groups = true;
            if (h->plan_source == 3 && pass_tune) {
                // the same check for every grouped launch the per-step stopwatch accepted
                materialise_groups();
                std::vector<std::vector<std::pair<int, int>>> accepted;          // per group: (op, selected plan)
                for (const auto& st : h->steps)
                    if (st.group >= 0) {
                        std::vector<std::pair<int, int>> mem;
                        const GroupLaunch& g = h->groups[st.group];
                        for (int m = 0; m < g.n_members; ++m) mem.push_back({g.op[m], gsel[g.op[m]]});
                        accepted.push_back(mem);
                    }
                float best = 0.f;
                rc = time_pass(&best); if (rc) return rc;
                for (const auto& mem : accepted) {
                    for (const auto& sv : mem) gsel[sv.first] = -1;
                    rc = build_steps(); if (rc) return rc;
                    materialise_groups();
                    float t = 0.f;
                    rc = time_pass(&t); if (rc) return rc;
                    const bool drop = t < 0.997f * best;
                    if (tune_log) fprintf(stderr, "[tune] pass check group with %s: %.1f us with -> %.1f us without: %s\n", h->convs[h->ops[mem[0].first].conv].name,
                                          best * 1e3, t * 1e3, drop ? "dropped" : "kept");
                    if (drop) best = t; else for (const auto& sv : mem) gsel[sv.first] = sv.second;
                }
                rc = build_steps(); if (rc) return rc;
            }
        }
        materialise_groups();
    }
    if (getenv("MI355_SCHED_LOG"))
        for (size_t k = 0; k < h->steps.size(); ++k) {
            fprintf(stderr, "[step] %zu: singles", k);
            for (int i : h->steps[k].singles) {
                fprintf(stderr, " %d:%s", i, h->ops[i].type == OP_CONV || h->ops[i].type == OP_STEM ? h->convs[h->ops[i].conv].name : h->ops[i].type == OP_UPSAMPLE ? "upsample" : "sppf_pools");
                if (h->ops[i].type == OP_CONV) fprintf(stderr, "[v%d,PT%d,CT%d,WP%d%s%s]", h->plans[i].version, h->plans[i].PT, h->plans[i].CT, h->plans[i].WP, h->plans[i].a.w2 ? ",+1x1" : "", h->plans[i].a.up_c ? ",up" : "");
            }
            if (h->steps[k].group >= 0) {
                fprintf(stderr, " | group");
                const GroupLaunch& g = h->groups[h->steps[k].group];
                for (int m = 0; m < g.n_members; ++m) fprintf(stderr, " %d:%s", g.op[m], h->convs[h->ops[g.op[m]].conv].name);
            }
            fprintf(stderr, "\n");
        }
    if (have) {
        std::vector<int> both(chosen);
        both.insert(both.end(), gsel.begin(), gsel.end());
        bool known = false;
        for (auto& t : h->tuned) if (t.first == shape_key) { t.second = both; known = true; }
        if (!known) h->tuned.push_back({shape_key, both});
        if (h->plan_source == 3) save_plan_choices(h, nb, Hl, Wl, n_cands, fp, chosen, gsel);
    }
    h->cur_nb = nb;
    h->plan_hash = fnv1a(fnv1a(fp, chosen.data(), chosen.size() * sizeof(int)), h->group_sel.data(), h->group_sel.size() * sizeof(int));   // candidates + choices: identifies the launch sequence
    h->plan_launches = 0;
    if (!h->steps.empty()) {
        for (const auto& st : h->steps) h->plan_launches += (int)st.singles.size() + (st.group >= 0);
    } else {
        for (size_t i = 0; i < h->ops.size(); ++i) h->plan_launches += launched((int)i);
    }
    if (getenv("MI355_SCHED_LOG")) {      // launch order of a pass for tools/layer_report.py: position, op index, stream, launched
        for (size_t pos = 0; pos < h->sched_order.size(); ++pos) {
            const int idx = h->sched_order[pos];
            fprintf(stderr, "[sched] %zu %d %d %d\n", pos, idx, h->op_stream[idx], !(h->ops[idx].type == OP_UPSAMPLE && h->fused_away[idx]) && !h->skip_op[idx]);
        }
    }
    return MI355_OK;
}

// run the net (+decode) on nb frames that sit in `frames_dev` (original size h0 x w0, dense).
// Launch-bound regime: the stem..decode sequence (60-100 launches) is captured once per chunk size into a hipGraph
// and replayed; the frames are first copied into the engine's own staging buffer so the captured pointers stay valid.
static int run_chunk(mi355_yolo* h, Prof& pf, const uint8_t* frames_dev, int nb, const Geometry& g, bool full_pred) {
    const uint8_t* stem_in = frames_dev;
    const bool graph = h->use_graph && !h->profiling && !full_pred;
    if (g.identity && graph) {
        HIPCHK(hipMemcpyAsync(h->lbox, frames_dev, (size_t)nb * g.Hl * g.Wl * 3, hipMemcpyDeviceToDevice, h->stream));
        stem_in = h->lbox;
    }
    if (!g.identity) {
        LetterboxArgs la{};
        la.src = frames_dev; la.H = g.h0; la.W = g.w0; la.frame_stride = (long long)g.h0 * g.w0 * 3; la.row_stride = g.w0 * 3;
        la.dst = h->lbox; la.Hd = g.Hl; la.Wd = g.Wl; la.top = g.top; la.left = g.left; la.Hr = g.Hr; la.Wr = g.Wr;
        la.xtab = h->d_xtab; la.ytab = h->d_ytab; la.resize = g.resize ? 1 : 0; la.B = nb;
        if (pf.begin(K_LETTERBOX)) return fail(MI355_EHIP, "event");
        KCHK(launch_letterbox(la, h->stream));
        pf.end();
        stem_in = h->lbox;
    }
    if (!graph) return launch_net(h, pf, stem_in, nb, g, full_pred);
    hipGraphExec_t exec = nullptr;
    for (auto& ge : h->graphs) if (ge.first == nb) exec = ge.second;
    if (!exec) {
        hipGraph_t gr = nullptr;
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        const int rc = launch_net(h, pf, stem_in, nb, g, full_pred);
        const hipError_t e = hipStreamEndCapture(h->stream, &gr);
        if (rc) { if (gr) (void)hipGraphDestroy(gr); return rc; }
        if (e != hipSuccess) return fail(MI355_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        const hipError_t ei = hipGraphInstantiate(&exec, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
        if (ei != hipSuccess) return fail(MI355_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
        h->graphs.push_back({nb, exec});
    }
    HIPCHK(hipGraphLaunch(exec, h->stream));
    return MI355_OK;
}

static int launch_net(mi355_yolo* h, Prof& pf, const uint8_t* stem_in, int nb, const Geometry& g, bool full_pred) {
    auto launch_op = [&](size_t i, hipStream_t st) -> int {
        const FileOp& o = h->ops[i];
        const int sd_out = h->bufs[o.dst_buf].stride_div;
        float* dst = h->view(o.dst_buf, o.dst_choff);
        if (o.type == OP_STEM) {
            const FileConv& c = h->convs[o.conv];
            StemArgs s{};
            s.img = stem_in; s.dst = dst; s.dst_cs = h->dbuf_cs[o.dst_buf];
            s.w = h->dconv[o.conv].w_raw; s.bias = h->dconv[o.conv].bias; s.lut = h->lut;
            s.B = nb; s.H = g.Hl; s.W = g.Wl; s.Hout = g.Hl / sd_out; s.Wout = g.Wl / sd_out;
            s.Cout = c.cout; s.k = c.k; s.stride = c.s; s.pad = c.pad;
            s.out_half = h->dbuf_es[o.dst_buf] == 2;
            if (pf.begin(K_STEM)) return fail(MI355_EHIP, "event");
            KCHK(launch_stem(s, st));
            pf.end();
        } else if (o.type == OP_CONV) {
            if (h->skip_op[i]) return MI355_OK;         // a pointwise conv that runs inside its producer's launch
            ConvLaunch l = h->plans[i];
            if (nb != h->cur_nb) {             // tail chunk: same buffers, fewer frames
                if (h->convs[o.conv].k == 1 && l.version == 3) {
                    const int sd_in = h->bufs[o.src_buf].stride_div;
                    l.a.Win = l.a.Wout = nb * (g.Hl / sd_in) * (g.Wl / sd_in);
                    const int per_block = 4 * 16 * (int)((size_t)l.a.TW / 64);       // TW = PT * 64 pixels per block
                    l.grid_x = (unsigned)((l.a.Wout + per_block - 1) / per_block);
                    if (pf.begin(K_CONV)) return fail(MI355_EHIP, "event");
                    KCHK(run_conv(l, st));
                    pf.end();
                    return MI355_OK;
                }
                if (h->convs[o.conv].k == 1) {
                    const int sd_in = h->bufs[o.src_buf].stride_div;
                    l.a.Win = l.a.Wout = nb * (g.Hl / sd_in) * (g.Wl / sd_in);
                    l.a.tiles_x = (l.a.Wout + l.a.TW - 1) / l.a.TW;
                    l.a.n_tiles_total = l.a.tiles_x;
                } else {
                    l.a.n_tiles_total = (int)((long)nb * l.a.tiles_x * l.a.tiles_y);
                }
                // v1: one block per tile; v4 (persistent): keep the planned grid unless fewer tiles exist
                l.grid_x = l.version == 4 ? std::min(l.grid_x, (unsigned)l.a.n_tiles_total) : (unsigned)l.a.n_tiles_total;
            }
            if (pf.begin(K_CONV)) return fail(MI355_EHIP, "event");
            KCHK(run_conv(l, st));
            pf.end();
        } else if (o.type == OP_UPSAMPLE) {
            if (h->fused_away[i]) return MI355_OK;      // read by its only consumer straight from the half-size map
            const int sd_in = h->bufs[o.src_buf].stride_div;
            if (pf.begin(K_UPSAMPLE)) return fail(MI355_EHIP, "event");
            if (h->dbuf_es[o.src_buf] != h->dbuf_es[o.dst_buf]) return fail(MI355_EFORMAT, "upsample between buffers of different precision");
            // fp16 buffers: a pure copy, so two halfs travel as one float (channel counts / offsets are multiples of 8)
            const int dv = h->dbuf_es[o.src_buf] == 2 ? 2 : 1;
            if (o.src_c % dv) return fail(MI355_EFORMAT, "half: odd channel count in upsample");
            KCHK(launch_upsample2x(h->view(o.src_buf, o.src_choff), h->dbuf_cs[o.src_buf] / dv, dst, h->dbuf_cs[o.dst_buf] / dv, nb,
                                   g.Hl / sd_in, g.Wl / sd_in, o.src_c / dv, st));
            pf.end();
        } else if (o.type == OP_SPPF_POOL) {
            if (o.k != 5) return fail(MI355_EFORMAT, "SPPF pool size must be 5");
            if (pf.begin(K_POOL)) return fail(MI355_EHIP, "event");
            if (h->dbuf_es[o.src_buf] == 2)
                KCHK(launch_sppf_pools_f16(h->view(o.src_buf, o.src_choff), h->dbuf_cs[o.src_buf], dst, h->dbuf_cs[o.dst_buf], nb,
                                           g.Hl / sd_out, g.Wl / sd_out, o.src_c, st));
            else
                KCHK(launch_sppf_pools(h->view(o.src_buf, o.src_choff), h->dbuf_cs[o.src_buf], dst, h->dbuf_cs[o.dst_buf], nb,
                                       g.Hl / sd_out, g.Wl / sd_out, o.src_c, st));
            pf.end();
        } else {
            return fail(MI355_EFORMAT, "unknown op type in program");
        }
        return MI355_OK;
    };
    // several streams along the dependency DAG (profiling keeps the single in-order stream; under hipGraph capture the
    // event waits fork the aux streams into the capture and the decode join brings them back)
    const bool multi = h->n_streams > 1 && !h->profiling && nb <= h->streams_max_batch && nb >= h->streams_min_batch;
    if (!multi && !h->steps.empty() && nb == h->cur_nb) {
        // single in-order stream, step by step: the ops of a step are mutually independent; its grouped convs are one grid
        for (const auto& stp : h->steps) {
            for (int i : stp.singles) { const int rc = launch_op((size_t)i, h->stream); if (rc) return rc; }
            if (stp.group >= 0) {
                if (pf.begin(K_CONV)) return fail(MI355_EHIP, "event");
                KCHK(run_group(h->groups[stp.group], h->stream));
                pf.end();
            }
        }
    } else if (!multi) {
        for (size_t i = 0; i < h->ops.size(); ++i) { const int rc = launch_op(i, h->stream); if (rc) return rc; }
    } else {
        for (int idx : h->sched_order) {
            const int sid = h->op_stream[idx];
            hipStream_t st = sid == 0 ? h->stream : h->aux[sid - 1];
            for (int dep : h->op_xdeps[idx]) {
                // an upsample fused into its consumer's read side is never launched (its event is never recorded): the
                // consumer already depends on the upsample's SOURCE producer (build_schedule)
                if (h->ops[dep].type == OP_UPSAMPLE && h->fused_away[dep]) continue;
                HIPCHK(hipStreamWaitEvent(st, h->op_done[dep], 0));
            }
            const int rc = launch_op((size_t)idx, st); if (rc) return rc;
            if (h->skip_op[idx]) continue;              // its event was recorded behind the producer's (fused) launch
            if (h->op_signals[idx] && !(h->ops[idx].type == OP_UPSAMPLE && h->fused_away[idx])) HIPCHK(hipEventRecord(h->op_done[idx], st));
            if (h->fuse2[idx] >= 0 && h->skip_op[h->fuse2[idx]] && h->op_signals[h->fuse2[idx]])
                HIPCHK(hipEventRecord(h->op_done[h->fuse2[idx]], st));
        }
        for (int l : h->leaf_ops)
            if (h->op_stream[l] != 0) HIPCHK(hipStreamWaitEvent(h->stream, h->op_done[l], 0));
    }
    DecodeArgs d{};
    d.n_levels = (int)h->levels.size();
    int a0 = 0;
    for (int l = 0; l < d.n_levels; ++l) {
        const FileLevel& lv = h->levels[l];
        d.lv[l] = HeadLevelArgs{h->dbuf[lv.buf], h->dbuf_cs[lv.buf], (int)lv.box_off, (int)lv.cls_off, (int)lv.kpt_off,
                                g.Hl / (int)lv.stride, g.Wl / (int)lv.stride, (int)lv.stride, a0};
        a0 += (g.Hl / lv.stride) * (g.Wl / lv.stride);
    }
    d.B = nb; d.A = h->A; d.nc = h->hdr.nc; d.nkpt = h->hdr.nkpt; d.kdim = h->hdr.kdim;
    d.pred = h->pred; d.best = h->best;
    if (pf.begin(K_DECODE)) return fail(MI355_EHIP, "event");
    KCHK(launch_decode(d, full_pred, h->stream));
    pf.end();
    return MI355_OK;
}

static int prepare_geometry(mi355_yolo* h, const Geometry& g, int imgsz) {
    if (g.resize && (h->tab_h0 != g.h0 || h->tab_w0 != g.w0 || h->tab_imgsz != imgsz)) {
        std::vector<int> xt, yt;
        resize_table(g.Wr, g.w0, xt); resize_table(g.Hr, g.h0, yt);
        if (h->d_xtab) (void)hipFree(h->d_xtab); if (h->d_ytab) (void)hipFree(h->d_ytab);
        h->d_xtab = h->d_ytab = nullptr;
        HIPCHK(hipMalloc(&h->d_xtab, xt.size() * 4)); HIPCHK(hipMalloc(&h->d_ytab, yt.size() * 4));
        HIPCHK(hipMemcpy(h->d_xtab, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_ytab, yt.data(), yt.size() * 4, hipMemcpyHostToDevice));
        h->tab_h0 = g.h0; h->tab_w0 = g.w0; h->tab_imgsz = imgsz;
    }
    return MI355_OK;
}

static int collect_timing(mi355_yolo* h, Prof& pf, int frames) {
    mi355_timing t{};
    t.frames = frames;
    (void)hipEventElapsedTime(&t.total_ms, h->ev0, h->ev1);
    for (auto& sp : pf.spans) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, h->pev[sp.second], h->pev[sp.second + 1]);
        switch (sp.first) {
            case K_LETTERBOX: t.letterbox_ms += ms; break;
            case K_STEM: t.stem_ms += ms; break;
            case K_CONV: t.conv_ms += ms; t.conv_launches++; break;
            case K_POOL: t.pool_ms += ms; break;
            case K_UPSAMPLE: t.upsample_ms += ms; break;
            case K_DECODE: t.decode_ms += ms; break;
            case K_NMS: t.nms_ms += ms; break;
        }
    }
    h->last = t;
    return MI355_OK;
}

// dev_rows != nullptr: the asynchronous device-output form (packed rows, counts and the row total stay in the caller's
// DEVICE buffers; nothing is copied to the host and the call returns with the work enqueued on the engine's stream)
static int infer_impl(mi355_yolo* h, const uint8_t* src, bool src_on_device, int n, int height, int width, int row_stride,
                      float conf, float iou, const int* classes, int n_classes, int max_det, int imgsz,
                      mi355_det* out_rows, int cap, int* out_counts, mi355_det* dev_rows = nullptr, int* dev_counts = nullptr,
                      int* dev_total = nullptr) {
    const bool async_out = dev_rows != nullptr;
    if (!h || !src || (!async_out && (!out_rows || !out_counts)) || (async_out && (!dev_counts || !dev_total)))
        return fail(MI355_EINVAL, "null argument");
    if (n <= 0 || height <= 0 || width <= 0) return fail(MI355_EINVAL, "n, height and width must be positive");
    if (max_det <= 0) max_det = 300;
    if (max_det > 1024) return fail(MI355_EINVAL, "max_det must be <= 1024");
    if (cap < 1) return fail(MI355_EINVAL, "out_capacity_per_image must be >= 1");
    if (imgsz <= 0) imgsz = 640;
    if (imgsz % 32) return fail(MI355_EINVAL, "imgsz must be a multiple of 32");
    if (row_stride == 0) row_stride = width * 3;
    if (row_stride < width * 3) return fail(MI355_EINVAL, "row_stride_bytes smaller than a row");
    if (n_classes < 0 || (n_classes > 0 && !classes)) return fail(MI355_EINVAL, "bad classes argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->async_pending) {             // an asynchronous call may still be reading the per-call scratch (class mask, row slots)
        HIPCHK(hipStreamSynchronize(h->stream));
        h->async_pending = false;
    }
    const Geometry g = make_geometry(height, width, imgsz);
    const int nb = std::min(n, h->chunk);
    int rc = ensure_shape(h, nb, g.Hl, g.Wl); if (rc) return rc;
    rc = prepare_geometry(h, g, imgsz); if (rc) return rc;

    const size_t frame_bytes = (size_t)height * width * 3;
    const uint8_t* dev_frames = src;
    // Host frames: a double-buffered staging area of two chunks.  Chunk k+1 is copied (on copy_stream) while chunk k's
    // kernels run; a slot is only overwritten after the kernels that read it (letterbox / stem) have been passed.
    auto copy_chunk = [&](int s0, int m, int slot) -> int {
        HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_consumed[slot], 0));
        HIPCHK(hipMemcpy2DAsync(h->d_in + (size_t)slot * nb * frame_bytes, (size_t)width * 3, src + (size_t)s0 * height * row_stride,
                                (size_t)row_stride, (size_t)width * 3, (size_t)height * m, hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(hipEventRecord(h->ev_copied[slot], h->copy_stream));
        return MI355_OK;
    };
    if (!src_on_device) {
        if (h->d_in_bytes < frame_bytes * nb * 2) {
            if (h->d_in) (void)hipFree(h->d_in);
            h->d_in = nullptr; h->d_in_bytes = 0;
            HIPCHK(hipMalloc(&h->d_in, frame_bytes * nb * 2)); h->d_in_bytes = frame_bytes * nb * 2;
        }
        HIPCHK(hipEventRecord(h->ev_consumed[0], h->stream));
        HIPCHK(hipEventRecord(h->ev_consumed[1], h->stream));
        rc = copy_chunk(0, std::min(nb, n), 0); if (rc) return rc;
    }
    if (h->rows_cap < (size_t)n * max_det) {
        if (h->d_rows) (void)hipFree(h->d_rows); h->d_rows = nullptr; h->rows_cap = 0;
        HIPCHK(hipMalloc(&h->d_rows, (size_t)n * max_det * sizeof(mi355_det))); h->rows_cap = (size_t)n * max_det;
    }
    if (h->counts_cap < 2 * n) {
        if (h->d_counts) (void)hipFree(h->d_counts); h->d_counts = nullptr; h->counts_cap = 0;
        HIPCHK(hipMalloc(&h->d_counts, (size_t)2 * n * sizeof(int) + 3 * h->chunk * sizeof(int))); h->counts_cap = 2 * n;   // + [counts | candidate counts | sort lengths] of one chunk
    }
    if (h->packed_cap < (size_t)n * max_det) {
        if (h->d_packed) (void)hipFree(h->d_packed); h->d_packed = nullptr; h->packed_cap = 0;
        HIPCHK(hipMalloc(&h->d_packed, (size_t)n * max_det * sizeof(mi355_det))); h->packed_cap = (size_t)n * max_det;
    }
    if (h->offsets_cap < n + 1) {
        if (h->d_offsets) (void)hipFree(h->d_offsets); h->d_offsets = nullptr; h->offsets_cap = 0;
        HIPCHK(hipMalloc(&h->d_offsets, (size_t)(n + 1) * sizeof(int))); h->offsets_cap = n + 1;
    }
    if (!async_out && h->h_rows_cap < (size_t)n * max_det) {
        if (h->h_rows) (void)hipHostFree(h->h_rows); h->h_rows = nullptr; h->h_rows_cap = 0;
        HIPCHK(hipHostMalloc(&h->h_rows, (size_t)n * max_det * sizeof(mi355_det))); h->h_rows_cap = (size_t)n * max_det;
    }
    if (h->h_counts_cap < n) {
        if (h->h_counts) (void)hipHostFree(h->h_counts); h->h_counts = nullptr; h->h_counts_cap = 0;
        HIPCHK(hipHostMalloc(&h->h_counts, (size_t)n * sizeof(int))); h->h_counts_cap = n;
    }
    const unsigned* cmask = nullptr;
    if (n_classes > 0) {
        const int words = ((int)h->hdr.nc + 31) / 32;
        if (h->cmask_words < words) {
            if (h->d_cmask) (void)hipFree(h->d_cmask); if (h->h_cmask) (void)hipHostFree(h->h_cmask);
            h->d_cmask = nullptr; h->h_cmask = nullptr; h->cmask_words = 0;
            HIPCHK(hipMalloc(&h->d_cmask, words * 4)); HIPCHK(hipHostMalloc(&h->h_cmask, words * 4)); h->cmask_words = words;
        }
        std::memset(h->h_cmask, 0, words * 4);
        for (int i = 0; i < n_classes; ++i)
            if (classes[i] >= 0 && classes[i] < (int)h->hdr.nc) h->h_cmask[classes[i] >> 5] |= 1u << (classes[i] & 31);
        HIPCHK(hipMemcpyAsync(h->d_cmask, h->h_cmask, words * 4, hipMemcpyHostToDevice, h->stream));
        cmask = h->d_cmask;
    }

    // Small synchronous calls (the reference's frame-by-frame loop, model.py:38): the greedy NMS kernel writes its rows and
    // counts straight into the pinned host buffers -- no compaction kernels, no copy-engine hand-overs (five stream operations,
    // ~45 us of a 425-us frame at batch 1), one stream synchronisation.  MI355_DIRECT_ROWS=0 keeps the copy path.
    const bool single_chunk = n <= nb;
    static const bool direct_rows_on = !(getenv("MI355_DIRECT_ROWS") && atoi(getenv("MI355_DIRECT_ROWS")) == 0);
    const bool direct_host = !async_out && single_chunk && n <= 16 && direct_rows_on;
    mi355_det* host_rows_dev = nullptr; int* host_counts_dev = nullptr;
    if (direct_host) {
        HIPCHK(hipHostGetDevicePointer((void**)&host_rows_dev, h->h_rows, 0));
        HIPCHK(hipHostGetDevicePointer((void**)&host_counts_dev, h->h_counts, 0));
    }
    Prof pf{h};
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int s = 0, ci = 0; s < n; s += nb, ++ci) {
        const int m = std::min(nb, n - s);
        const uint8_t* chunk_frames = dev_frames + (size_t)s * frame_bytes;
        if (!src_on_device) {
            const int slot = ci & 1;
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_copied[slot], 0));
            chunk_frames = h->d_in + (size_t)slot * nb * frame_bytes;
        }
        rc = run_chunk(h, pf, chunk_frames, m, g, false); if (rc) return rc;
        if (!src_on_device) {
            // the frames of this slot have been consumed once the net's kernels are enqueued behind this event; the
            // (host-blocking) copy of the next chunk is issued AFTER this chunk's launches so that it overlaps them
            HIPCHK(hipEventRecord(h->ev_consumed[ci & 1], h->stream));
            if (s + nb < n) { rc = copy_chunk(s + nb, std::min(nb, n - s - nb), (ci + 1) & 1); if (rc) return rc; }
        }
        NmsArgs na{};
        na.pred = h->pred; na.best = h->best; na.B = m; na.A = h->A; na.no = h->no(); na.nc = h->hdr.nc;
        na.nk = h->hdr.nkpt * h->hdr.kdim; na.kdim = h->hdr.kdim;
        na.conf = conf; na.iou = iou; na.max_det = max_det; na.max_nms = 30000; na.max_wh = 7680.f;
        na.class_mask = cmask; na.keys = h->keys; na.Apow2 = h->Apow2;
        na.scale_back = 1; na.gain = (float)g.gain; na.pad_x = (float)g.pad_x; na.pad_y = (float)g.pad_y;
        na.kpad_x = (float)g.kpad_x; na.kpad_y = (float)g.kpad_y; na.orig_w = (float)width; na.orig_h = (float)height;
        na.out_rows = h->d_rows + (size_t)s * max_det;
        if (direct_host) {                       // rows and counts straight into the pinned host buffers (slot layout: frame i at i * max_det)
            na.out_rows = host_rows_dev;
            na.host_counts = host_counts_dev;
        }
        if (pf.begin(K_NMS)) return fail(MI355_EHIP, "event");
        if (single_chunk) {
            // one chunk: the sort kernels' scratch [n, 3n) lies inside the counts allocation (2n + 3 * chunk ints, n <= chunk)
            na.out_counts = h->d_counts;
            KCHK(launch_nms(na, h->stream));
        } else {
            // counts of this chunk belong at [s, s + m), but the sort kernels use out_counts[B, 3B) as scratch: they run on a
            // temporary block [2n, 2n + 3 * chunk) and the counts are copied into place
            int* tmp = h->d_counts + 2 * n;
            NmsArgs nb_args = na; nb_args.out_counts = tmp;
            KCHK(launch_nms(nb_args, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_counts + s, tmp, (size_t)m * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        }
        pf.end();
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    if (async_out) {
        // packed rows (frame order), counts and their sum go to the caller's device buffers; no host copy, no sync
        KCHK(launch_compact_rows(h->d_rows, h->d_counts, n, max_det, (int)(sizeof(mi355_det) / 4), h->d_offsets, dev_rows, h->stream));
        HIPCHK(hipMemcpyAsync(dev_counts, h->d_counts, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(dev_total, h->d_offsets + n, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
        h->async_pending = true;
        return MI355_OK;
    }
    if (direct_host) {
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int i = 0; i < n; ++i) {
            const int c = std::min(h->h_counts[i], cap);
            out_counts[i] = c;
            std::memcpy(out_rows + (size_t)i * cap, h->h_rows + (size_t)i * max_det, (size_t)c * sizeof(mi355_det));
        }
        return collect_timing(h, pf, n);
    }
    // rows -> host: compact on the GPU first (a frame keeps counts[i] of its max_det slots; copying the slots would be 35 MB
    // per 512 frames), then two small copies: the counts, and sum(counts) rows
    KCHK(launch_compact_rows(h->d_rows, h->d_counts, n, max_det, (int)(sizeof(mi355_det) / 4), h->d_offsets, h->d_packed, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_counts, h->d_counts, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    // Small calls (the reference's frame-by-frame loop): the first rows travel speculatively behind the counts, so that one
    // stream synchronisation serves both copies (a sync costs 15-20 us; at batch 1 the whole frame takes 500); a second
    // copy follows only when a call keeps more rows than were guessed.
    const size_t guess = n <= 16 ? std::min((size_t)n * max_det, (size_t)64 * n) : 0;
    if (guess) HIPCHK(hipMemcpyAsync(h->h_rows, h->d_packed, guess * sizeof(mi355_det), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (size_t)h->h_counts[i];
    if (total > guess) {
        HIPCHK(hipMemcpyAsync(h->h_rows + guess, h->d_packed + guess, (total - guess) * sizeof(mi355_det), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    size_t at = 0;
    for (int i = 0; i < n; ++i) {
        const int c = std::min(h->h_counts[i], cap);
        out_counts[i] = c;
        std::memcpy(out_rows + (size_t)i * cap, h->h_rows + at, (size_t)c * sizeof(mi355_det));
        at += (size_t)h->h_counts[i];
    }
    return collect_timing(h, pf, n);
}

static int create_impl(const uint8_t* blob, size_t nbytes, int device_id, const mi355_opts* opts, mi355_yolo** out) {
    if (!blob || !out) return fail(MI355_EINVAL, "null argument");
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(MI355_EINVAL, "device_id out of range (no MI355X visible?)");
    HIPCHK(hipSetDevice(device_id));
    std::unique_ptr<mi355_yolo> h(new mi355_yolo());
    h->device = device_id;
    if (opts && opts->struct_size >= (int)sizeof(mi355_opts) && opts->batch_chunk > 0) h->chunk = opts->batch_chunk;
    if (opts && opts->struct_size >= (int)sizeof(mi355_opts)) h->half = opts->half != 0;
    if (h->half) h->autotune = 28;     // the fp16 plan space also spans the pixel tiles per wave
    if (const char* e = getenv("MI355_AUTOTUNE")) h->autotune = std::max(0, atoi(e));
    if (const char* e = getenv("MI355_GRAPH")) h->use_graph = atoi(e);
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&h->ev0)); HIPCHK(hipEventCreate(&h->ev1));
    HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_consumed[i], hipEventDisableTiming));
    }
    const int rc = parse_blob(h.get(), blob, nbytes);
    if (rc) return rc;
    *out = h.release();
    return MI355_OK;
}

struct DevMem {   // RAII for the one-shot operator entry points
    std::vector<void*> ptrs;
    ~DevMem() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) ptrs.push_back(*p); return e; }
};

}  // namespace mi355

// ================================================================================================ C ABI
extern "C" {

const char* mi355_last_error(void) { return g_err.c_str(); }

int mi355_yolo_create_from_memory(const void* blob, size_t nbytes, int device_id, const mi355_opts* opts, mi355_yolo** out) {
    return create_impl((const uint8_t*)blob, nbytes, device_id, opts, out);
}

int mi355_yolo_create(const char* path, int device_id, const mi355_opts* opts, mi355_yolo** out) {
    if (!path || !out) return fail(MI355_EINVAL, "null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(MI355_EIO, std::string("cannot open weights file: ") + path);
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(sz > 0 ? (size_t)sz : 0);
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size() || buf.empty()) return fail(MI355_EIO, std::string("cannot read weights file: ") + path);
    return create_impl(buf.data(), buf.size(), device_id, opts, out);
}

void mi355_yolo_destroy(mi355_yolo* h) { delete h; }

int mi355_yolo_info(const mi355_yolo* h, mi355_model_info* info) {
    if (!h || !info) return fail(MI355_EINVAL, "null argument");
    std::memset(info, 0, sizeof(*info));
    info->task = h->hdr.task; info->nc = h->hdr.nc; info->nkpt = h->hdr.nkpt; info->kdim = h->hdr.kdim;
    info->reg_max = h->hdr.reg_max; info->n_levels = (int)h->levels.size();
    for (size_t i = 0; i < h->levels.size(); ++i) info->strides[i] = h->levels[i].stride;
    info->n_convs = (int)h->convs.size(); info->n_ops = (int)h->ops.size(); info->n_buffers = (int)h->bufs.size();
    info->n_params = h->n_params; info->macs_640 = h->macs640;
    std::strncpy(info->family, h->hdr.family == 0 ? "v8" : "v5u", sizeof(info->family) - 1);
    info->scale = (char)h->hdr.scale;
    return MI355_OK;
}

int mi355_yolo_infer(mi355_yolo* h, const uint8_t* bgr, int n, int height, int width, int row_stride, float conf, float iou,
                     const int* classes, int n_classes, int max_det, int imgsz, mi355_det* out_rows, int cap, int* out_counts) {
    return infer_impl(h, bgr, false, n, height, width, row_stride, conf, iou, classes, n_classes, max_det, imgsz, out_rows, cap, out_counts);
}

int mi355_yolo_infer_device(mi355_yolo* h, const uint8_t* bgr_dev, int n, int height, int width, float conf, float iou,
                            const int* classes, int n_classes, int max_det, int imgsz, mi355_det* out_rows, int cap, int* out_counts) {
    return infer_impl(h, bgr_dev, true, n, height, width, 0, conf, iou, classes, n_classes, max_det, imgsz, out_rows, cap, out_counts);
}

int mi355_yolo_infer_device_async(mi355_yolo* h, const uint8_t* bgr_dev, int n, int height, int width, float conf, float iou,
                                  const int* classes, int n_classes, int max_det, int imgsz, mi355_det* rows_dev, int* counts_dev,
                                  int* total_dev) {
    if (!rows_dev) return fail(MI355_EINVAL, "null argument");
    return infer_impl(h, bgr_dev, true, n, height, width, 0, conf, iou, classes, n_classes, max_det, imgsz, nullptr, 1, nullptr,
                      rows_dev, counts_dev, total_dev);
}

void* mi355_yolo_stream(mi355_yolo* h) { return h ? (void*)h->stream : nullptr; }

int mi355_yolo_sync(mi355_yolo* h) {
    if (!h) return fail(MI355_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->async_pending = false;
    return MI355_OK;
}

int mi355_yolo_plan_info(const mi355_yolo* h, unsigned long long* plan_hash, int* source, int* launches, long long* activation_bytes,
                         long long* activation_bytes_unshared) {
    if (!h) return fail(MI355_EINVAL, "null argument");
    if (plan_hash) *plan_hash = h->plan_hash;
    if (source) *source = h->plan_source;
    if (launches) *launches = h->plan_launches;
    if (activation_bytes) *activation_bytes = h->act_bytes;
    if (activation_bytes_unshared) *activation_bytes_unshared = h->act_bytes_noreuse;
    return MI355_OK;
}

int mi355_memory_plan(const void* blob, size_t nbytes, int n, int height, int width, int imgsz, int half, int reuse, long long* offsets,
                      long long* sizes, int cap, int* n_buffers, long long* arena_bytes, long long* unshared_bytes) {
    if (!blob || n <= 0 || height <= 0 || width <= 0 || cap < 0 || (cap > 0 && (!offsets || !sizes))) return fail(MI355_EINVAL, "bad argument");
    if (imgsz <= 0) imgsz = 640;
    if (imgsz % 32) return fail(MI355_EINVAL, "imgsz must be a multiple of 32");
    mi355_yolo h;
    h.host_only = true; h.half = half != 0;
    const int rc = parse_blob(&h, (const uint8_t*)blob, nbytes); if (rc) return rc;
    h.mem_reuse = reuse;
    const Geometry g = make_geometry(height, width, imgsz);
    std::vector<size_t> off, bytes; size_t arena = 0, plain = 0;
    plan_memory(&h, n, g.Hl, g.Wl, &off, &bytes, &arena, &plain);
    if (n_buffers) *n_buffers = (int)off.size();
    for (size_t i = 0; i < off.size() && (int)i < cap; ++i) { offsets[i] = (long long)off[i]; sizes[i] = (long long)bytes[i]; }
    if (arena_bytes) *arena_bytes = (long long)arena;
    if (unshared_bytes) *unshared_bytes = (long long)plain;
    return MI355_OK;
}

int mi355_yolo_set_profiling(mi355_yolo* h, int on) {
    if (!h) return fail(MI355_EINVAL, "null argument");
    h->profiling = on != 0;
    return MI355_OK;
}

int mi355_yolo_last_timing(const mi355_yolo* h, mi355_timing* t) {
    if (!h || !t) return fail(MI355_EINVAL, "null argument");
    *t = h->last;
    return MI355_OK;
}

int mi355_yolo_raw_head(mi355_yolo* h, const uint8_t* bgr, int n, int height, int width, int row_stride, int imgsz,
                        float* out, int* out_channels, int* out_anchors) {
    if (!h || !out_channels || !out_anchors) return fail(MI355_EINVAL, "null argument");
    if (n <= 0 || height <= 0 || width <= 0) return fail(MI355_EINVAL, "n, height and width must be positive");
    if (imgsz <= 0) imgsz = 640;
    if (imgsz % 32) return fail(MI355_EINVAL, "imgsz must be a multiple of 32");
    const Geometry g = make_geometry(height, width, imgsz);
    int A = 0;
    for (const FileLevel& lv : h->levels) A += (g.Hl / lv.stride) * (g.Wl / lv.stride);
    *out_channels = h->no(); *out_anchors = A;
    if (!out) return MI355_OK;
    if (!bgr) return fail(MI355_EINVAL, "null argument");
    if (row_stride == 0) row_stride = width * 3;
    HIPCHK(hipSetDevice(h->device));
    const int nb = std::min(n, h->chunk);
    int rc = ensure_shape(h, nb, g.Hl, g.Wl); if (rc) return rc;
    rc = prepare_geometry(h, g, imgsz); if (rc) return rc;
    const size_t frame_bytes = (size_t)height * width * 3;
    if (h->d_in_bytes < frame_bytes * n) {
        if (h->d_in) (void)hipFree(h->d_in);
        h->d_in = nullptr; h->d_in_bytes = 0;
        HIPCHK(hipMalloc(&h->d_in, frame_bytes * n)); h->d_in_bytes = frame_bytes * n;
    }
    HIPCHK(hipMemcpy2DAsync(h->d_in, (size_t)width * 3, bgr, (size_t)row_stride, (size_t)width * 3, (size_t)height * n,
                            hipMemcpyHostToDevice, h->stream));
    const size_t per = (size_t)A * h->no();
    if (h->rawhead_floats < per * nb) {
        if (h->d_rawhead) (void)hipFree(h->d_rawhead); h->d_rawhead = nullptr; h->rawhead_floats = 0;
        HIPCHK(hipMalloc(&h->d_rawhead, per * nb * 4)); h->rawhead_floats = per * nb;
    }
    Prof pf{h};
    const bool was = h->profiling; h->profiling = false;
    for (int s = 0; s < n; s += nb) {
        const int m = std::min(nb, n - s);
        rc = run_chunk(h, pf, h->d_in + (size_t)s * frame_bytes, m, g, true);
        if (rc) { h->profiling = was; return rc; }
        KCHK(launch_transpose_pred(h->pred, h->d_rawhead, m, A, h->no(), h->stream));
        HIPCHK(hipMemcpyAsync(out + (size_t)s * per, h->d_rawhead, per * m * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->profiling = was;
    return MI355_OK;
}

// ------------------------------------------------------------------------------------- single operators
int mi355_letterbox_shape(int height, int width, int imgsz, int* out_h, int* out_w) {
    if (!out_h || !out_w || height <= 0 || width <= 0 || imgsz <= 0) return fail(MI355_EINVAL, "bad argument");
    const Geometry g = make_geometry(height, width, imgsz);
    *out_h = g.Hl; *out_w = g.Wl;
    return MI355_OK;
}


int mi355_op_letterbox(int device_id, const uint8_t* bgr, int n, int height, int width, int imgsz, uint8_t* out) {
    if (!bgr || !out || n <= 0 || height <= 0 || width <= 0 || imgsz <= 0) return fail(MI355_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(device_id));
    const Geometry g = make_geometry(height, width, imgsz);
    DevMem dm; uint8_t *d_src, *d_dst; int *d_x, *d_y;
    std::vector<int> xt, yt;
    resize_table(g.Wr, g.w0, xt); resize_table(g.Hr, g.h0, yt);
    const size_t sb = (size_t)n * height * width * 3, db = (size_t)n * g.Hl * g.Wl * 3;
    HIPCHK(dm.alloc(&d_src, sb)); HIPCHK(dm.alloc(&d_dst, db)); HIPCHK(dm.alloc(&d_x, xt.size() * 4)); HIPCHK(dm.alloc(&d_y, yt.size() * 4));
    HIPCHK(hipMemcpy(d_src, bgr, sb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_x, xt.data(), xt.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_y, yt.data(), yt.size() * 4, hipMemcpyHostToDevice));
    LetterboxArgs la{};
    la.src = d_src; la.H = height; la.W = width; la.frame_stride = (long long)height * width * 3; la.row_stride = width * 3;
    la.dst = d_dst; la.Hd = g.Hl; la.Wd = g.Wl; la.top = g.top; la.left = g.left; la.Hr = g.Hr; la.Wr = g.Wr;
    la.xtab = d_x; la.ytab = d_y; la.resize = g.resize ? 1 : 0; la.B = n;
    KCHK(launch_letterbox(la, nullptr));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_dst, db, hipMemcpyDeviceToHost));
    return MI355_OK;
}

static int op_conv2d_impl(int device_id, const float* x, int n, int h, int w, int cin, const float* w_oihw, const float* bias,
                          int cout, int k, int stride, int silu, const float* residual, float* y, int plan_index, int* n_plans,
                          bool half, bool out_f32) {
    if (!x || !w_oihw || !bias || !y || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return fail(MI355_EINVAL, "bad argument");
    if (!((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)))) return fail(MI355_EINVAL, "k/stride not supported");
    if ((h % stride) || (w % stride)) return fail(MI355_EINVAL, "h and w must be multiples of the stride");
    HIPCHK(hipSetDevice(device_id));
    const int ho = h / stride, wo = w / stride;
    const int es_in = half ? 2 : 4, es_out = (half && !out_f32) ? 2 : 4;
    const int cs_in = round_up(cin, 16 / es_in), cs_out = round_up(cout, 16 / es_out);
    const size_t npi = (size_t)n * h * w, npo = (size_t)n * ho * wo;
    // host images of the padded NHWC tensors, in the device dtype (fp32 -> fp16 is round-to-nearest-even)
    std::vector<float> xin(npi * cs_in, 0.f), yout(npo * cs_out, 0.f), rs;
    for (size_t p = 0; p < npi; ++p) std::memcpy(&xin[p * cs_in], x + p * cin, (size_t)cin * 4);
    auto upload = [&](float** dptr, DevMem& dm, const std::vector<float>& v, int es) -> int {
        HIPCHK(dm.alloc(dptr, v.size() * es));
        if (es == 4) { HIPCHK(hipMemcpy(*dptr, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return MI355_OK; }
        std::vector<uint16_t> hb(v.size());
        floats_to_halfs(v.data(), hb.data(), v.size());
        HIPCHK(hipMemcpy(*dptr, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        return MI355_OK;
    };
    DevMem dm; float *d_x, *d_y, *d_r = nullptr, *d_w, *d_b, *d_z;
    HIPCHK(dm.alloc(&d_z, 256)); HIPCHK(hipMemset(d_z, 0, 256));
    int rc = upload(&d_x, dm, xin, es_in); if (rc) return rc;
    HIPCHK(dm.alloc(&d_y, yout.size() * es_out));
    HIPCHK(hipMemset(d_y, 0, yout.size() * es_out));
    if (residual) {
        rs.assign(npo * cs_out, 0.f);
        for (size_t p = 0; p < npo; ++p) std::memcpy(&rs[p * cs_out], residual + p * cout, (size_t)cout * 4);
        rc = upload(&d_r, dm, rs, es_out); if (rc) return rc;
    }
    std::vector<float> bp(round_up(cout, 16), 0.f);
    std::memcpy(bp.data(), bias, (size_t)cout * 4);
    HIPCHK(dm.alloc(&d_b, bp.size() * 4));
    HIPCHK(hipMemcpy(d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    if (half) {
        std::vector<uint16_t> pk(packed_weight_halfs(cout, cin, k));
        pack_conv_weights_f16(w_oihw, cout, cin, k, pk.data());
        HIPCHK(dm.alloc(&d_w, pk.size() * 2));
        HIPCHK(hipMemcpy(d_w, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    } else {
        std::vector<float> pk(packed_weight_floats(cout, cin, k));
        pack_conv_weights(w_oihw, cout, cin, k, pk.data());
        HIPCHK(dm.alloc(&d_w, pk.size() * 4));
        HIPCHK(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    }
    ConvArgs a{};
    a.src = d_x; a.src_cs = cs_in; a.dst = d_y; a.dst_cs = cs_out; a.res = d_r; a.res_cs = cs_out; a.wpk = d_w; a.bias = d_b;
    a.B = n; a.Hin = h; a.Win = w; a.Hout = ho; a.Wout = wo; a.Cin = cin; a.Cout = cout; a.k = k; a.stride = stride; a.pad = k / 2; a.act = silu ? 1 : 0;
    a.zeros = d_z; a.dtype = half ? 1 : 0; a.out_f32 = (half && out_f32) ? 1 : 0;
    std::vector<ConvLaunch> cands;
    KCHK(plan_conv_candidates(a, &cands));
    // plan_index: which candidate launch plan to run (tests sweep it to cover every kernel variant and wave shape)
    const ConvLaunch& l = cands[(size_t)(plan_index < 0 ? 0 : plan_index) % cands.size()];
    if (n_plans) *n_plans = (int)cands.size();
    KCHK(run_conv(l, nullptr));
    HIPCHK(hipDeviceSynchronize());
    if (es_out == 4) {
        HIPCHK(hipMemcpy(yout.data(), d_y, yout.size() * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> hb(yout.size());
        HIPCHK(hipMemcpy(hb.data(), d_y, hb.size() * 2, hipMemcpyDeviceToHost));
        halfs_to_floats(hb.data(), yout.data(), hb.size());
    }
    for (size_t p = 0; p < npo; ++p) std::memcpy(y + p * cout, &yout[p * cs_out], (size_t)cout * 4);
    return MI355_OK;
}

int mi355_op_conv2d(int device_id, const float* x, int n, int h, int w, int cin, const float* w_oihw, const float* bias,
                    int cout, int k, int stride, int silu, const float* residual, float* y, int plan_index, int* n_plans) {
    return op_conv2d_impl(device_id, x, n, h, w, cin, w_oihw, bias, cout, k, stride, silu, residual, y, plan_index, n_plans, false, false);
}

int mi355_op_conv2d_f16(int device_id, const float* x, int n, int h, int w, int cin, const float* w_oihw, const float* bias,
                        int cout, int k, int stride, int silu, const float* residual, float* y, int out_f32, int plan_index,
                        int* n_plans) {
    return op_conv2d_impl(device_id, x, n, h, w, cin, w_oihw, bias, cout, k, stride, silu, residual, y, plan_index, n_plans, true,
                          out_f32 != 0);
}

// Conv3x3 (+bias+SiLU) -> Conv1x1 (+bias, optional SiLU) as ONE fused launch (conv_igemm_f32 / _f16 <..., F2 = true>): the parity
// hook of the fused pairs the engine runs (stride-2 conv -> C2f.cv1, head branch [1] -> [2]).
static int op_conv2d_fused_impl(int device_id, const float* x, int n, int h, int w, int cin, const float* w1_oihw, const float* b1, int c1,
                                int stride, const float* w2_oihw, const float* b2, int c2, int silu2, float* y, int plan_index, int* n_plans,
                                bool half, bool out_f32, const float* residual = nullptr, const float* lead = nullptr, int lead_c = 0) {
    if (!x || !w1_oihw || !b1 || !w2_oihw || !b2 || !y || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || c1 <= 0 || c2 <= 0) return fail(MI355_EINVAL, "bad argument");
    if ((stride != 1 && stride != 2) || (h % stride) || (w % stride)) return fail(MI355_EINVAL, "stride must be 1 or 2 and divide h and w");
    HIPCHK(hipSetDevice(device_id));
    const int ho = h / stride, wo = w / stride;
    const int es_in = half ? 2 : 4, es_out = (half && !out_f32) ? 2 : 4;
    const int cs_in = round_up(cin, 16 / es_in), cs_out = round_up(c2, 16 / es_out), cs_mid = round_up(c1, 16 / es_in);
    const size_t npi = (size_t)n * h * w, npo = (size_t)n * ho * wo;
    std::vector<float> xin(npi * cs_in, 0.f), yout(npo * cs_out, 0.f);
    for (size_t p = 0; p < npi; ++p) std::memcpy(&xin[p * cs_in], x + p * cin, (size_t)cin * 4);
    DevMem dm; float *d_x, *d_y, *d_w1, *d_b1, *d_w2, *d_b2, *d_z, *d_mid;
    HIPCHK(dm.alloc(&d_z, 256)); HIPCHK(hipMemset(d_z, 0, 256));
    HIPCHK(dm.alloc(&d_x, xin.size() * es_in));
    if (half) {
        std::vector<uint16_t> hb(xin.size());
        floats_to_halfs(xin.data(), hb.data(), xin.size());
        HIPCHK(hipMemcpy(d_x, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    } else {
        HIPCHK(hipMemcpy(d_x, xin.data(), xin.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(dm.alloc(&d_y, yout.size() * es_out)); HIPCHK(hipMemset(d_y, 0, yout.size() * es_out));
    HIPCHK(dm.alloc(&d_mid, npo * cs_mid * es_in));               // the unfused destination of the first conv: must stay untouched
    HIPCHK(hipMemset(d_mid, 0, npo * cs_mid * es_in));
    auto upload_conv = [&](const float* wt, const float* b, int co, int ci, int k, float** dw, float** db) -> int {
        std::vector<float> bp(round_up(co, 16), 0.f);
        std::memcpy(bp.data(), b, (size_t)co * 4);
        if (half) {
            std::vector<uint16_t> pk(packed_weight_halfs(co, ci, k));
            pack_conv_weights_f16(wt, co, ci, k, pk.data());
            HIPCHK(dm.alloc(dw, pk.size() * 2)); HIPCHK(hipMemcpy(*dw, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        } else {
            std::vector<float> pk(packed_weight_floats(co, ci, k));
            pack_conv_weights(wt, co, ci, k, pk.data());
            HIPCHK(dm.alloc(dw, pk.size() * 4)); HIPCHK(hipMemcpy(*dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        }
        HIPCHK(dm.alloc(db, bp.size() * 4)); HIPCHK(hipMemcpy(*db, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
        return MI355_OK;
    };
    if ((residual || lead_c) && half) return fail(MI355_EINVAL, "residual / lead channels exist in the fp32 fused kernel only");
    if (lead_c < 0 || (lead_c > 0 && !lead)) return fail(MI355_EINVAL, "bad lead argument");
    int rc = upload_conv(w1_oihw, b1, c1, cin, 3, &d_w1, &d_b1); if (rc) return rc;
    rc = upload_conv(w2_oihw, b2, c2, lead_c + c1, 1, &d_w2, &d_b2); if (rc) return rc;       // pointwise weights over cat(lead, conv1 output)
    ConvArgs a{};
    // lead channels and the first conv's (unused) destination share ONE buffer [lead | mid], as the C2f concat buffer does
    float* d_cat = nullptr; float* d_res = nullptr;
    const int cs_cat = round_up(lead_c + c1, 4);
    if (lead_c) {
        std::vector<float> cat(npo * cs_cat, 0.f);
        for (size_t p = 0; p < npo; ++p) std::memcpy(&cat[p * cs_cat], lead + p * lead_c, (size_t)lead_c * 4);
        HIPCHK(dm.alloc(&d_cat, cat.size() * 4)); HIPCHK(hipMemcpy(d_cat, cat.data(), cat.size() * 4, hipMemcpyHostToDevice));
    }
    if (residual) {
        std::vector<float> rs(npo * cs_mid, 0.f);
        for (size_t p = 0; p < npo; ++p) std::memcpy(&rs[p * cs_mid], residual + p * c1, (size_t)c1 * 4);
        HIPCHK(dm.alloc(&d_res, rs.size() * 4)); HIPCHK(hipMemcpy(d_res, rs.data(), rs.size() * 4, hipMemcpyHostToDevice));
        a.res = d_res; a.res_cs = cs_mid;
    }
    if (lead_c) { a.f2_lead = d_cat; a.f2_lead_cs = cs_cat; a.f2_lead_c = lead_c; }
    a.src = d_x; a.src_cs = cs_in; a.dst = lead_c ? d_cat + lead_c : d_mid; a.dst_cs = lead_c ? cs_cat : cs_mid; a.wpk = d_w1; a.bias = d_b1; a.zeros = d_z;
    a.B = n; a.Hin = h; a.Win = w; a.Hout = ho; a.Wout = wo; a.Cin = cin; a.Cout = c1; a.k = 3; a.stride = stride; a.pad = 1; a.act = 1;
    a.dtype = half ? 1 : 0;
    a.f2_wpk = d_w2; a.f2_bias = d_b2; a.f2_dst = d_y; a.f2_dst_cs = cs_out; a.f2_cout = c2; a.f2_act = silu2 ? 1 : 0;
    a.f2_out_f32 = (half && out_f32) ? 1 : 0;
    std::vector<ConvLaunch> cands;
    KCHK(plan_conv_candidates(a, &cands));
    if (n_plans) *n_plans = (int)cands.size();
    KCHK(run_conv(cands[(size_t)(plan_index < 0 ? 0 : plan_index) % cands.size()], nullptr));
    HIPCHK(hipDeviceSynchronize());
    if (es_out == 4) {
        HIPCHK(hipMemcpy(yout.data(), d_y, yout.size() * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> hb(yout.size());
        HIPCHK(hipMemcpy(hb.data(), d_y, hb.size() * 2, hipMemcpyDeviceToHost));
        halfs_to_floats(hb.data(), yout.data(), hb.size());
    }
    for (size_t p = 0; p < npo; ++p) std::memcpy(y + p * c2, &yout[p * cs_out], (size_t)c2 * 4);
    return MI355_OK;
}

// Two independent convs (same input tensor, different weights) run as ONE grouped launch (conv_f32_group.hip) with candidate
// plans plan_a / plan_b (indices into each conv's candidate list, skipping plans whose kernel is not on the group kernel's
// menu: *n_menu_a / *n_menu_b return how many are): the parity hook of the grouped launches -- must equal mi355_op_conv2d of each.
int mi355_op_conv2d_group(int device_id, const float* x, int n, int h, int w, int cin, const float* wa, const float* ba, int cout_a, int k_a,
                          int stride_a, const float* wb, const float* bb, int cout_b, int k_b, int stride_b, float* ya, float* yb, int plan_a,
                          int plan_b, int* n_menu_a, int* n_menu_b, const float* w2a, const float* b2a, int cout2_a) {
    if (!x || !wa || !ba || !wb || !bb || !ya || !yb || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout_a <= 0 || cout_b <= 0) return fail(MI355_EINVAL, "bad argument");
    if (cout2_a > 0 && (!w2a || !b2a || k_a != 3)) return fail(MI355_EINVAL, "the fused pointwise stage needs weights and a 3x3 first conv");
    HIPCHK(hipSetDevice(device_id));
    const int cs_in = round_up(cin, 4);
    const size_t npi = (size_t)n * h * w;
    std::vector<float> xin(npi * cs_in, 0.f);
    for (size_t p = 0; p < npi; ++p) std::memcpy(&xin[p * cs_in], x + p * cin, (size_t)cin * 4);
    DevMem dm; float *d_x, *d_z;
    HIPCHK(dm.alloc(&d_z, 256)); HIPCHK(hipMemset(d_z, 0, 256));
    HIPCHK(dm.alloc(&d_x, xin.size() * 4)); HIPCHK(hipMemcpy(d_x, xin.data(), xin.size() * 4, hipMemcpyHostToDevice));
    struct One { const float* w; const float* b; int cout, k, stride; float* y; int cout2; float* d_y; int cs_out; size_t npo; std::vector<ConvLaunch> menu; std::vector<int> kinds; };
    One c[2] = {{wa, ba, cout_a, k_a, stride_a, ya, cout2_a > 0 ? cout2_a : 0}, {wb, bb, cout_b, k_b, stride_b, yb, 0}};
    for (One& o : c) {
        if (!((o.k == 1 && o.stride == 1) || (o.k == 3 && (o.stride == 1 || o.stride == 2))) || (h % o.stride) || (w % o.stride)) return fail(MI355_EINVAL, "k/stride not supported");
        const int c_final = o.cout2 ? o.cout2 : o.cout;                 // channels of the tensor that is written
        o.cs_out = round_up(c_final, 4); o.npo = (size_t)n * (h / o.stride) * (w / o.stride);
        float *d_w, *d_b;
        std::vector<float> pk(packed_weight_floats(o.cout, cin, o.k)), bp(round_up(o.cout, 16), 0.f);
        pack_conv_weights(o.w, o.cout, cin, o.k, pk.data());
        std::memcpy(bp.data(), o.b, (size_t)o.cout * 4);
        HIPCHK(dm.alloc(&d_w, pk.size() * 4)); HIPCHK(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(dm.alloc(&d_b, bp.size() * 4)); HIPCHK(hipMemcpy(d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(dm.alloc(&o.d_y, o.npo * o.cs_out * 4)); HIPCHK(hipMemset(o.d_y, 0, o.npo * o.cs_out * 4));
        ConvArgs a{};
        a.src = d_x; a.src_cs = cs_in; a.dst = o.d_y; a.dst_cs = o.cs_out; a.wpk = d_w; a.bias = d_b; a.zeros = d_z;
        a.B = n; a.Hin = h; a.Win = w; a.Hout = h / o.stride; a.Wout = w / o.stride; a.Cin = cin; a.Cout = o.cout; a.k = o.k; a.stride = o.stride;
        a.pad = o.k / 2; a.act = 1;
        if (o.cout2) {                       // Conv3x3 -> Conv1x1 fused: the 3x3's own output goes nowhere, the 1x1 writes d_y
            float *d_mid, *d_w2, *d_b2;
            HIPCHK(dm.alloc(&d_mid, o.npo * round_up(o.cout, 4) * 4));
            std::vector<float> pk2(packed_weight_floats(o.cout2, o.cout, 1)), bp2(round_up(o.cout2, 16), 0.f);
            pack_conv_weights(w2a, o.cout2, o.cout, 1, pk2.data());
            std::memcpy(bp2.data(), b2a, (size_t)o.cout2 * 4);
            HIPCHK(dm.alloc(&d_w2, pk2.size() * 4)); HIPCHK(hipMemcpy(d_w2, pk2.data(), pk2.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(dm.alloc(&d_b2, bp2.size() * 4)); HIPCHK(hipMemcpy(d_b2, bp2.data(), bp2.size() * 4, hipMemcpyHostToDevice));
            a.dst = d_mid; a.dst_cs = round_up(o.cout, 4);
            a.f2_wpk = d_w2; a.f2_bias = d_b2; a.f2_dst = o.d_y; a.f2_dst_cs = o.cs_out; a.f2_cout = o.cout2; a.f2_act = 0;
        }
        std::vector<ConvLaunch> cands;
        KCHK(plan_conv_candidates(a, &cands));
        for (const ConvLaunch& l : cands) { const int kd = group_kind(l, o.k, o.stride); if (kd >= 0) { o.menu.push_back(l); o.kinds.push_back(kd); } }
    }
    if (n_menu_a) *n_menu_a = (int)c[0].menu.size();
    if (n_menu_b) *n_menu_b = (int)c[1].menu.size();
    if (c[0].menu.empty() || c[1].menu.empty()) return fail(MI355_EINVAL, "no candidate plan of one conv is on the group kernel's menu");
    const size_t ia = (size_t)(plan_a < 0 ? 0 : plan_a) % c[0].menu.size(), ib = (size_t)(plan_b < 0 ? 0 : plan_b) % c[1].menu.size();
    GroupLaunch g{};
    KCHK(plan_group({c[0].menu[ia], c[1].menu[ib]}, {c[0].kinds[ia], c[1].kinds[ib]}, &g));
    KCHK(run_group(g, nullptr));
    HIPCHK(hipDeviceSynchronize());
    for (One& o : c) {
        std::vector<float> yo(o.npo * o.cs_out);
        HIPCHK(hipMemcpy(yo.data(), o.d_y, yo.size() * 4, hipMemcpyDeviceToHost));
        const int c_final = o.cout2 ? o.cout2 : o.cout;
        for (size_t p = 0; p < o.npo; ++p) std::memcpy(o.y + p * c_final, &yo[p * o.cs_out], (size_t)c_final * 4);
    }
    return MI355_OK;
}

int mi355_op_conv2d_fused(int device_id, const float* x, int n, int h, int w, int cin, const float* w1_oihw, const float* b1, int c1,
                          int stride, const float* w2_oihw, const float* b2, int c2, int silu2, float* y, int plan_index, int* n_plans) {
    return op_conv2d_fused_impl(device_id, x, n, h, w, cin, w1_oihw, b1, c1, stride, w2_oihw, b2, c2, silu2, y, plan_index, n_plans, false, false);
}

int mi355_op_c2f_tail(int device_id, const float* x, int n, int h, int w, int cin, const float* w1_oihw, const float* b1, int c1,
                      const float* residual, const float* lead, int lead_c, const float* w2_oihw, const float* b2, int c2, float* y,
                      int plan_index, int* n_plans) {
    return op_conv2d_fused_impl(device_id, x, n, h, w, cin, w1_oihw, b1, c1, 1, w2_oihw, b2, c2, 1, y, plan_index, n_plans, false, false,
                                residual, lead, lead_c);
}

int mi355_op_conv2d_fused_f16(int device_id, const float* x, int n, int h, int w, int cin, const float* w1_oihw, const float* b1, int c1,
                              int stride, const float* w2_oihw, const float* b2, int c2, int silu2, float* y, int out_f32, int plan_index,
                              int* n_plans) {
    return op_conv2d_fused_impl(device_id, x, n, h, w, cin, w1_oihw, b1, c1, stride, w2_oihw, b2, c2, silu2, y, plan_index, n_plans, true,
                                out_f32 != 0);
}

static int bench_conv2d_impl(int device_id, int n, int h, int w, int cin, int cout, int k, int stride, int silu, int residual,
                             int plan_index, int iters, float* avg_ms, int* n_plans, char* plan_desc, int plan_desc_len, bool half) {
    if (!avg_ms || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || iters <= 0) return fail(MI355_EINVAL, "bad argument");
    if (!((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)))) return fail(MI355_EINVAL, "k/stride not supported");
    HIPCHK(hipSetDevice(device_id));
    const int es = half ? 2 : 4;
    const int ho = h / stride, wo = w / stride, cs_in = round_up(cin, 16 / es), cs_out = round_up(cout, 16 / es);
    const size_t nin = (size_t)n * h * w * cs_in, nout = (size_t)n * ho * wo * cs_out;     // elements
    DevMem dm; float *d_x, *d_y, *d_r = nullptr, *d_w, *d_b, *d_z;
    HIPCHK(dm.alloc(&d_x, nin * es)); HIPCHK(dm.alloc(&d_y, nout * es)); HIPCHK(dm.alloc(&d_z, 256)); HIPCHK(hipMemset(d_z, 0, 256));
    {   // random activations / weights (benchmarks on zeros read high: DVFS)
        std::vector<float> hx(std::min<size_t>(nin, 1u << 22));
        std::vector<uint16_t> hh(half ? hx.size() : 0);
        unsigned st = 12345u;
        for (float& v : hx) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
        if (half) floats_to_halfs(hx.data(), hh.data(), hx.size());
        const void* hsrc = half ? (const void*)hh.data() : (const void*)hx.data();
        for (size_t o = 0; o < nin; o += hx.size())
            HIPCHK(hipMemcpy((char*)d_x + o * es, hsrc, std::min(hx.size(), nin - o) * es, hipMemcpyHostToDevice));
        if (residual) { HIPCHK(dm.alloc(&d_r, nout * es)); HIPCHK(hipMemcpy(d_r, d_x, std::min(nin, nout) * es, hipMemcpyDeviceToDevice)); }
        std::vector<float> wt((size_t)cout * cin * k * k), bp(round_up(cout, 16), 0.1f);
        for (float& v : wt) { st = st * 1664525u + 1013904223u; v = (((st >> 8) & 0xffff) / 32768.0f - 1.0f) / std::sqrt((float)cin * k * k); }
        if (half) {
            std::vector<uint16_t> pk(packed_weight_halfs(cout, cin, k));
            pack_conv_weights_f16(wt.data(), cout, cin, k, pk.data());
            HIPCHK(dm.alloc(&d_w, pk.size() * 2));
            HIPCHK(hipMemcpy(d_w, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        } else {
            std::vector<float> pk(packed_weight_floats(cout, cin, k));
            pack_conv_weights(wt.data(), cout, cin, k, pk.data());
            HIPCHK(dm.alloc(&d_w, pk.size() * 4));
            HIPCHK(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        }
        HIPCHK(dm.alloc(&d_b, bp.size() * 4));
        HIPCHK(hipMemcpy(d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    }
    ConvArgs a{};
    a.src = d_x; a.src_cs = cs_in; a.dst = d_y; a.dst_cs = cs_out; a.res = d_r; a.res_cs = cs_out; a.wpk = d_w; a.bias = d_b; a.zeros = d_z;
    a.dtype = half ? 1 : 0;
    a.B = n; a.Hin = h; a.Win = w; a.Hout = ho; a.Wout = wo; a.Cin = cin; a.Cout = cout; a.k = k; a.stride = stride; a.pad = k / 2; a.act = silu ? 1 : 0;
    std::vector<ConvLaunch> cands;
    KCHK(plan_conv_candidates(a, &cands));
    if (n_plans) *n_plans = (int)cands.size();
    const ConvLaunch& l = cands[(size_t)(plan_index < 0 ? 0 : plan_index) % cands.size()];
    if (plan_desc && plan_desc_len > 0)
        snprintf(plan_desc, plan_desc_len, "v%d CT%d PT%d WP%d tile %dx%d ck%d lds %zu grid %ux%u", l.version, l.CT, l.PT, l.WP, l.a.TW, l.a.TH, l.a.ck,
                 l.lds, l.grid_x, l.grid_y);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) KCHK(run_conv(l, nullptr));
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) KCHK(run_conv(l, nullptr));
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = ms / iters;
    return MI355_OK;
}

int mi355_bench_conv2d(int device_id, int n, int h, int w, int cin, int cout, int k, int stride, int silu, int residual,
                       int plan_index, int iters, float* avg_ms, int* n_plans, char* plan_desc, int plan_desc_len) {
    return bench_conv2d_impl(device_id, n, h, w, cin, cout, k, stride, silu, residual, plan_index, iters, avg_ms, n_plans, plan_desc,
                             plan_desc_len, false);
}

int mi355_bench_conv2d_f16(int device_id, int n, int h, int w, int cin, int cout, int k, int stride, int silu, int residual,
                           int plan_index, int iters, float* avg_ms, int* n_plans, char* plan_desc, int plan_desc_len) {
    return bench_conv2d_impl(device_id, n, h, w, cin, cout, k, stride, silu, residual, plan_index, iters, avg_ms, n_plans, plan_desc,
                             plan_desc_len, true);
}

// Host-only view of the launch planner (no kernel is launched, no device memory is touched): which kernel versions would be
// offered for a conv of this shape and these buffer strides.  Used by the CPU tests of the planner's guards.
int mi355_plan_query(int n, int h, int w, int cin, int cout, int k, int stride, int src_cs, int dst_cs, int res_cs, int f2_cout,
                     int f2_dst_cs, int half, int* versions, int cap, int* n_plans) {
    if (!n_plans || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || cap < 0 || (cap > 0 && !versions)) return fail(MI355_EINVAL, "bad argument");
    if (stride != 1 && stride != 2) return fail(MI355_EINVAL, "stride must be 1 or 2");
    float* fake = (float*)(uintptr_t)0x10000;                    // aligned, never dereferenced
    ConvArgs a{};
    a.src = fake; a.src_cs = src_cs; a.dst = fake; a.dst_cs = dst_cs; a.res = res_cs ? fake : nullptr; a.res_cs = res_cs;
    a.wpk = fake; a.bias = fake; a.zeros = fake;
    a.B = n; a.Hin = h; a.Win = w; a.Hout = h / stride; a.Wout = w / stride; a.Cin = cin; a.Cout = cout; a.k = k; a.stride = stride;
    a.pad = k / 2; a.act = 1; a.dtype = half ? 1 : 0;
    if (f2_cout > 0) { a.f2_wpk = fake; a.f2_bias = fake; a.f2_dst = fake; a.f2_dst_cs = f2_dst_cs; a.f2_cout = f2_cout; a.f2_act = 0; }
    std::vector<ConvLaunch> cands;
    if (const char* e = plan_conv_candidates(a, &cands)) { *n_plans = 0; return fail(MI355_EINVAL, e); }
    *n_plans = (int)cands.size();
    for (int i = 0; i < (int)cands.size() && i < cap; ++i) versions[i] = cands[i].version + (cands[i].a.w2 ? 100 : 0);
    return MI355_OK;
}

int mi355_op_stem(int device_id, const uint8_t* bgr, int n, int h, int w, const float* w_oihw, const float* bias, int cout,
                  int k, int stride, float* y) {
    if (!bgr || !w_oihw || !bias || !y || n <= 0 || h <= 0 || w <= 0 || cout <= 0) return fail(MI355_EINVAL, "bad argument");
    if ((k != 3 && k != 6) || (h % stride) || (w % stride)) return fail(MI355_EINVAL, "k/stride not supported");
    HIPCHK(hipSetDevice(device_id));
    const int ho = h / stride, wo = w / stride, cs = round_up(cout, 4);
    DevMem dm; uint8_t* d_img; float *d_y, *d_w, *d_b, *d_l;
    const size_t ib = (size_t)n * h * w * 3, yn = (size_t)n * ho * wo * cs;
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (float)i / 255.0f;
    HIPCHK(dm.alloc(&d_img, ib)); HIPCHK(dm.alloc(&d_y, yn * 4)); HIPCHK(dm.alloc(&d_w, (size_t)cout * 3 * k * k * 4));
    HIPCHK(dm.alloc(&d_b, (size_t)cout * 4)); HIPCHK(dm.alloc(&d_l, sizeof(lut)));
    HIPCHK(hipMemcpy(d_img, bgr, ib, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_w, w_oihw, (size_t)cout * 3 * k * k * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_b, bias, (size_t)cout * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_l, lut, sizeof(lut), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_y, 0, yn * 4));
    StemArgs s{};
    s.img = d_img; s.dst = d_y; s.dst_cs = cs; s.w = d_w; s.bias = d_b; s.lut = d_l;
    s.B = n; s.H = h; s.W = w; s.Hout = ho; s.Wout = wo; s.Cout = cout; s.k = k; s.stride = stride; s.pad = (k == 6 ? 2 : k / 2);
    KCHK(launch_stem(s, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> yo(yn);
    HIPCHK(hipMemcpy(yo.data(), d_y, yn * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < (size_t)n * ho * wo; ++p) std::memcpy(y + p * cout, &yo[p * cs], (size_t)cout * 4);
    return MI355_OK;
}

int mi355_op_nms(int device_id, const float* pred, int n, int nc, int extra, int anchors, float conf, float iou,
                 const int* classes, int n_classes, int max_det, mi355_det* out_rows, int cap, int* out_counts) {
    if (!pred || !out_rows || !out_counts || n <= 0 || nc <= 0 || extra < 0 || anchors <= 0 || cap < 1) return fail(MI355_EINVAL, "bad argument");
    if (max_det <= 0) max_det = 300;
    if (max_det > 1024) return fail(MI355_EINVAL, "max_det must be <= 1024");
    if (extra > MI355_MAX_KPT_FLOATS) return fail(MI355_EINVAL, "too many extra columns");
    HIPCHK(hipSetDevice(device_id));
    const int no = 4 + nc + extra;
    int ap2 = 1; while (ap2 < anchors) ap2 <<= 1;
    DevMem dm; float *d_in, *d_am; float2* d_best; unsigned long long* d_keys; mi355_det* d_rows; int* d_counts; unsigned* d_mask = nullptr;
    const size_t pn = (size_t)n * no * anchors;
    HIPCHK(dm.alloc(&d_in, pn * 4)); HIPCHK(dm.alloc(&d_am, pn * 4)); HIPCHK(dm.alloc(&d_best, (size_t)n * anchors * sizeof(float2)));
    HIPCHK(dm.alloc(&d_keys, (size_t)n * ap2 * 8)); HIPCHK(dm.alloc(&d_rows, (size_t)n * max_det * sizeof(mi355_det)));
    HIPCHK(dm.alloc(&d_counts, (size_t)3 * n * sizeof(int)));
    HIPCHK(hipMemcpy(d_in, pred, pn * 4, hipMemcpyHostToDevice));
    if (n_classes > 0 && classes) {
        std::vector<unsigned> m((nc + 31) / 32, 0u);
        for (int i = 0; i < n_classes; ++i) if (classes[i] >= 0 && classes[i] < nc) m[classes[i] >> 5] |= 1u << (classes[i] & 31);
        HIPCHK(dm.alloc(&d_mask, m.size() * 4));
        HIPCHK(hipMemcpy(d_mask, m.data(), m.size() * 4, hipMemcpyHostToDevice));
    }
    KCHK(launch_transpose_pred(d_in, d_am, n, no, anchors, nullptr));      // [n][no][A] -> [n][A][no]
    KCHK(launch_best_from_pred(d_am, n, anchors, no, nc, d_best, nullptr));
    NmsArgs na{};
    na.pred = d_am; na.best = d_best; na.B = n; na.A = anchors; na.no = no; na.nc = nc; na.nk = extra; na.kdim = 0;
    na.conf = conf; na.iou = iou; na.max_det = max_det; na.max_nms = 30000; na.max_wh = 7680.f;
    na.class_mask = d_mask; na.keys = d_keys; na.Apow2 = ap2; na.scale_back = 0; na.gain = 1.f;
    na.out_rows = d_rows; na.out_counts = d_counts;
    KCHK(launch_nms(na, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<mi355_det> rows((size_t)n * max_det);
    std::vector<int> counts(n);
    HIPCHK(hipMemcpy(rows.data(), d_rows, rows.size() * sizeof(mi355_det), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(counts.data(), d_counts, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        const int c = std::min(counts[i], cap);
        out_counts[i] = c;
        std::memcpy(out_rows + (size_t)i * cap, rows.data() + (size_t)i * max_det, (size_t)c * sizeof(mi355_det));
    }
    return MI355_OK;
}

}  // extern "C"
