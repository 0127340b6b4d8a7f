// Where the activation buffers live: ONE arena, buffers whose lifetimes cannot overlap under any RAW-respecting schedule share bytes.
#include "engine_internal.h"

namespace mi355 {

// Liveness-based placement of the activation buffers in ONE arena (host arithmetic only).  Fills h->dbuf_cs / dbuf_es.
void plan_memory(mi355_yolo* h, int nb, int Hl, int Wl, std::vector<size_t>* off_out, std::vector<size_t>* bytes_out,
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

}  // namespace mi355
