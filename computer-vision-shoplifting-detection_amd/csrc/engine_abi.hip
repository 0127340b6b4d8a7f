// The C ABI of include/mi355_yolo.h: engine handle entry points (create / infer / raw_head / plan_info / memory_plan ...).
#include "engine_internal.h"

namespace mi355 { thread_local std::string g_err; }
using namespace mi355;

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

}  // extern "C"
