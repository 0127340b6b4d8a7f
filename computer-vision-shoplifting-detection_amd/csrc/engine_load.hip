// .mi355w image -> engine object: program validation, weight upload (MFMA fragment order), the fusable patterns of the
// program (upsample on read, Conv3x3 -> Conv1x1, C2f tail), the dependency DAG and its stream assignment.
// Replaces ultralytics Model.__init__ / AutoBackend (weights + fuse) behind /root/reference/model.py:18.
#include "engine_internal.h"

using namespace mi355;

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
    for (auto& c : dconv) { if (c.wpk) (void)hipFree(c.wpk); if (c.bias) (void)hipFree(c.bias); if (c.w_raw) (void)hipFree(c.w_raw); if (c.w_frag) (void)hipFree(c.w_frag); }
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

int parse_blob(mi355_yolo* h, const uint8_t* blob, size_t n) {
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
            if (h->half && c.k == 3) {         // A fragments of stem3s2_u8_h
                std::vector<uint16_t> fr;
                stem3_weight_frags(w, (int)c.cout, fr);
                HIPCHK(hipMalloc(&d.w_frag, fr.size() * 2));
                HIPCHK(hipMemcpy(d.w_frag, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
            } else if (c.k == 3) {             // ... of stem3s2_u8_f32
                std::vector<float> fr;
                stem3_weight_frags_f32(w, (int)c.cout, fr);
                HIPCHK(hipMalloc(&d.w_frag, fr.size() * 4));
                HIPCHK(hipMemcpy(d.w_frag, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
            }
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
    if (fenv ? atoi(fenv) != 0 : !(h->opt_flags & MI355_OPT_NO_FUSE_UPSAMPLE)) {
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
    const bool fuse_tail = (f3env ? atoi(f3env) != 0 : !(h->opt_flags & MI355_OPT_NO_FUSE_TAIL)) && !h->half;      // fp32 kernels only
    if (f2env ? atoi(f2env) != 0 : !(h->opt_flags & MI355_OPT_NO_FUSE_1X1)) {
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

int create_impl(const uint8_t* blob, size_t nbytes, int device_id, const mi355_opts* opts, mi355_yolo** out) {
    if (!blob || !out) return fail(MI355_EINVAL, "null argument");
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(MI355_EINVAL, "device_id out of range (no MI355X visible?)");
    HIPCHK(hipSetDevice(device_id));
    std::unique_ptr<mi355_yolo> h(new mi355_yolo());
    h->device = device_id;
    // options: a field is read only when the caller's struct holds it (struct_size), so callers built against an earlier header work
    auto has = [&](size_t off, size_t sz) { return opts && opts->struct_size >= (int)(off + sz); };
#define MI355_HAS(f) has(offsetof(mi355_opts, f), sizeof(opts->f))
    if (MI355_HAS(batch_chunk) && opts->batch_chunk > 0) h->chunk = opts->batch_chunk;
    if (MI355_HAS(half)) h->half = opts->half != 0;
    if (MI355_HAS(fast_act)) h->fast_act = opts->fast_act != 0 && !h->half;
    if (h->half) h->autotune = 28;     // the fp16 plan space also spans the pixel tiles per wave
    if (MI355_HAS(autotune) && opts->autotune != 0) h->autotune = std::max(0, opts->autotune);
    if (MI355_HAS(streams) && opts->streams > 0) h->n_streams = std::min(8, opts->streams);
    if (MI355_HAS(flags)) h->opt_flags = opts->flags;
    if (MI355_HAS(plan_dir) && opts->plan_dir) h->plan_dir = opts->plan_dir;
    if (const char* home = getenv("HOME")) h->plan_cache_dir = std::string(home) + "/.cache/mi355yolo"; else h->plan_cache_dir = "/tmp/mi355yolo";
    if (MI355_HAS(plan_cache_dir) && opts->plan_cache_dir) { h->plan_cache_dir = opts->plan_cache_dir; h->plan_cache_on = !h->plan_cache_dir.empty(); }
#undef MI355_HAS
    h->use_groups = (h->opt_flags & MI355_OPT_NO_GROUPS) ? 0 : 1;
    h->mem_reuse = (h->opt_flags & MI355_OPT_NO_MEM_REUSE) ? 0 : 1;
    h->use_graph = (h->opt_flags & MI355_OPT_HIP_GRAPH) ? 1 : 0;
    // A/B overrides (tools/*.sh): never needed to configure the product
    if (const char* e = getenv("MI355_AUTOTUNE")) h->autotune = std::max(0, atoi(e));
    if (const char* e = getenv("MI355_GRAPH")) h->use_graph = atoi(e);
    if (const char* e = getenv("MI355_FAST_ACT")) h->fast_act = atoi(e) != 0 && !h->half;
    if (const char* e = getenv("MI355_PLAN_DIR")) h->plan_dir = e;
    if (const char* e = getenv("MI355_PLAN_CACHE")) { h->plan_cache_on = std::strcmp(e, "0") != 0 && *e; if (h->plan_cache_on) h->plan_cache_dir = e; }
    {
        // default priority; MI355_ENGINE_PRIO=1 asks for the highest the device offers (A/B only: measured no gain for the detector when
        // the tracker's motion compensation shares the GPU, tools/track_prio_ab.sh)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        static const bool high_prio = getenv("MI355_ENGINE_PRIO") && atoi(getenv("MI355_ENGINE_PRIO")) == 1;
        HIPCHK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, high_prio ? greatest : 0));
    }
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

}  // namespace mi355
