// Launch plans of one shape (frames per pass, letterboxed H x W): candidate lists from the planner (conv_plan.hip), choices from
// this process's memory / a plan file / the stopwatch, the step schedule and grouped launches of the latency-bound regime.
#include "engine_internal.h"

namespace mi355 {

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

static std::string plan_file_name(const mi355_yolo* h, int nb, int Hl, int Wl) {
    char name[160];
    snprintf(name, sizeof(name), "/%016llx_%s_t%d_%dx%dx%d.plan", h->model_hash, h->half ? "f16" : h->fast_act ? "f32fast" : "f32", h->autotune, nb, Hl, Wl);
    return name;
}
// where freshly timed choices are written (and found again); "" = not persisted
static std::string plan_cache_path(const mi355_yolo* h, int nb, int Hl, int Wl) {
    if (!h->plan_cache_on || h->plan_cache_dir.empty()) return std::string();
    const std::string& dir = h->plan_cache_dir;
    (void)mkdir(dir.substr(0, dir.rfind('/')).c_str(), 0755);
    (void)mkdir(dir.c_str(), 0755);
    return dir + plan_file_name(h, nb, Hl, Wl);
}

constexpr int kUpBase = 10000;       // chosen[i] >= kUpBase: the conv reads the upsample kernel's output with plan chosen[i] - kUpBase

static bool load_plan_file(const std::string& path, const std::vector<int>& n_cands, unsigned long long fp, std::vector<int>* chosen, std::vector<int>* gsel) {
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

// The shipped directory (opts.plan_dir: the choices of the benchmarked workloads, committed with the package) comes first, so that every
// process on the same GPU model launches the same sequence; then this machine's cache.  -> 0 none, 1 shipped, 2 cache
static int load_plan_choices(const mi355_yolo* h, int nb, int Hl, int Wl, const std::vector<int>& n_cands, unsigned long long fp,
                             std::vector<int>* chosen, std::vector<int>* gsel) {
    if (!h->plan_dir.empty() && load_plan_file(h->plan_dir + plan_file_name(h, nb, Hl, Wl), n_cands, fp, chosen, gsel)) return 1;
    if (load_plan_file(plan_cache_path(h, nb, Hl, Wl), n_cands, fp, chosen, gsel)) return 2;
    return 0;
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

int ensure_shape(mi355_yolo* h, int nb, int Hl, int Wl) {
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
        a.Cin = c.cin; a.Cout = c.cout; a.k = c.k; a.stride = c.s; a.pad = c.pad; a.act = (c.act && h->fast_act) ? 2 : c.act;
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
            f.f2_dst = h->view(o1.dst_buf, o1.dst_choff); f.f2_dst_cs = h->dbuf_cs[o1.dst_buf]; f.f2_cout = c1.cout; f.f2_act = (c1.act && h->fast_act) ? 2 : c1.act;
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
    if (!have && h->autotune) {
        const int from = load_plan_choices(h, nb, Hl, Wl, n_cands, fp, &chosen, &gsel);
        if (from) { have = have_groups = true; h->plan_source = from == 1 ? 4 : 2; }
    }
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
        const int pass_tune = getenv("MI355_PASS_TUNE") ? atoi(getenv("MI355_PASS_TUNE")) : !(h->opt_flags & MI355_OPT_NO_PASS_TUNE);
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

}  // namespace mi355
