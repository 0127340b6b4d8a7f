// One pass of the net (launch_net), one chunk (run_chunk) and one infer call (infer_impl): letterbox -> stem -> conv program ->
// decode -> NMS -> rows.  Replaces BasePredictor.stream_inference behind /root/reference/model.py:38.
#include "engine_internal.h"

namespace mi355 {

// run the net (+decode) on nb frames that sit in `frames_dev` (original size h0 x w0, dense).
// Launch-bound regime: the stem..decode sequence (60-100 launches) is captured once per chunk size into a hipGraph
// and replayed; the frames are first copied into the engine's own staging buffer so the captured pointers stay valid.
int run_chunk(mi355_yolo* h, Prof& pf, const uint8_t* frames_dev, int nb, const Geometry& g, bool full_pred) {
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

int launch_net(mi355_yolo* h, Prof& pf, const uint8_t* stem_in, int nb, const Geometry& g, bool full_pred) {
    auto launch_op = [&](size_t i, hipStream_t st) -> int {
        const FileOp& o = h->ops[i];
        const int sd_out = h->bufs[o.dst_buf].stride_div;
        float* dst = h->view(o.dst_buf, o.dst_choff);
        if (o.type == OP_STEM) {
            const FileConv& c = h->convs[o.conv];
            StemArgs s{};
            s.img = stem_in; s.dst = dst; s.dst_cs = h->dbuf_cs[o.dst_buf];
            s.w = h->dconv[o.conv].w_raw; s.bias = h->dconv[o.conv].bias; s.lut = h->lut; s.wfrag = h->dconv[o.conv].w_frag;
            s.B = nb; s.H = g.Hl; s.W = g.Wl; s.Hout = g.Hl / sd_out; s.Wout = g.Wl / sd_out;
            s.Cout = c.cout; s.k = c.k; s.stride = c.s; s.pad = c.pad;
            s.out_half = h->dbuf_es[o.dst_buf] == 2; s.fast_act = h->fast_act ? 1 : 0;
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
                // v7 (persistent over (tile, cout group) units): likewise
                l.grid_x = l.version == 4 ? std::min(l.grid_x, (unsigned)l.a.n_tiles_total)
                         : (l.version == 7 || l.version == 10) ? std::min(l.grid_x, (unsigned)l.a.n_tiles_total * (unsigned)l.a.cgroups) : (unsigned)l.a.n_tiles_total;
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

int prepare_geometry(mi355_yolo* h, const Geometry& g, int imgsz) {
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
int infer_impl(mi355_yolo* h, const uint8_t* src, bool src_on_device, int n, int height, int width, int row_stride,
                      float conf, float iou, const int* classes, int n_classes, int max_det, int imgsz,
                      mi355_det* out_rows, int cap, int* out_counts, mi355_det* dev_rows, int* dev_counts, int* dev_total) {
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
    const bool direct_rows_on = getenv("MI355_DIRECT_ROWS") ? atoi(getenv("MI355_DIRECT_ROWS")) != 0 : !(h->opt_flags & MI355_OPT_NO_DIRECT_ROWS);
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

}  // namespace mi355
