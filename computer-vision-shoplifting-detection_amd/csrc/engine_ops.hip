// Single-operator entry points of include/mi355_yolo.h (host pointers in and out; the parity tests isolate a kernel with them).
#include <limits>
#include "engine_internal.h"

using namespace mi355;

extern "C" {

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

// Pointwise conv over cat(upsample2x(x_half), x_skip) with the upsample fused into the conv's read side (fp32): the parity hook of the
// neck's Upsample -> Concat -> C2f.cv1 chain as the engine runs it (the up channels of the concat buffer are never written: they
// are poisoned here, so a plan that reads them shows).
static int op_conv1x1_upcat_impl(int device_id, const float* x_half, const float* x_skip, int n, int h, int w, int up_c, int skip_c,
                                 const float* w_oihw, const float* bias, int cout, int silu, float* y, int plan_index, int* n_plans, bool half) {
    if (!x_half || !x_skip || !w_oihw || !bias || !y || n <= 0 || h <= 0 || w <= 0 || up_c <= 0 || skip_c <= 0 || cout <= 0) return fail(MI355_EINVAL, "bad argument");
    if ((h & 1) || (w & 1) || (up_c & 15)) return fail(MI355_EINVAL, "h and w must be even and up_c a multiple of 16");
    HIPCHK(hipSetDevice(device_id));
    const int es = half ? 2 : 4, al = 16 / es;
    const int cin = up_c + skip_c, cs_in = round_up(cin, al), cs_h = round_up(up_c, al), cs_out = round_up(cout, al);
    const size_t np = (size_t)n * h * w, nph = np / 4;
    std::vector<float> xin(np * cs_in, 0.f), xh(nph * cs_h, 0.f), yout(np * cs_out, 0.f);
    const float poison = std::numeric_limits<float>::quiet_NaN();
    for (size_t p = 0; p < np; ++p) {
        for (int c = 0; c < up_c; ++c) xin[p * cs_in + c] = poison;
        std::memcpy(&xin[p * cs_in + up_c], x_skip + p * skip_c, (size_t)skip_c * 4);
    }
    for (size_t p = 0; p < nph; ++p) std::memcpy(&xh[p * cs_h], x_half + p * up_c, (size_t)up_c * 4);
    DevMem dm; float *d_x, *d_h, *d_y, *d_w, *d_b, *d_z;
    HIPCHK(dm.alloc(&d_z, 256)); HIPCHK(hipMemset(d_z, 0, 256));
    auto upload = [&](float** dptr, const std::vector<float>& v) -> int {
        HIPCHK(dm.alloc(dptr, v.size() * es));
        if (!half) { HIPCHK(hipMemcpy(*dptr, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return MI355_OK; }
        std::vector<uint16_t> hb(v.size());
        floats_to_halfs(v.data(), hb.data(), v.size());
        HIPCHK(hipMemcpy(*dptr, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
        return MI355_OK;
    };
    int rc = upload(&d_x, xin); if (rc) return rc;
    rc = upload(&d_h, xh); if (rc) return rc;
    HIPCHK(dm.alloc(&d_y, yout.size() * es)); HIPCHK(hipMemset(d_y, 0, yout.size() * es));
    std::vector<float> bp(round_up(cout, 16), 0.f);
    std::memcpy(bp.data(), bias, (size_t)cout * 4);
    if (half) {
        std::vector<uint16_t> pk(packed_weight_halfs(cout, cin, 1));
        pack_conv_weights_f16(w_oihw, cout, cin, 1, pk.data());
        HIPCHK(dm.alloc(&d_w, pk.size() * 2)); HIPCHK(hipMemcpy(d_w, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    } else {
        std::vector<float> pk(packed_weight_floats(cout, cin, 1));
        pack_conv_weights(w_oihw, cout, cin, 1, pk.data());
        HIPCHK(dm.alloc(&d_w, pk.size() * 4)); HIPCHK(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(dm.alloc(&d_b, bp.size() * 4)); HIPCHK(hipMemcpy(d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a{};
    a.src = d_x; a.src_cs = cs_in; a.dst = d_y; a.dst_cs = cs_out; a.wpk = d_w; a.bias = d_b; a.zeros = d_z;
    a.src2 = d_h; a.src2_cs = cs_h; a.up_c = up_c; a.dtype = half ? 1 : 0;
    a.B = n; a.Hin = h; a.Win = w; a.Hout = h; a.Wout = w; a.Cin = cin; a.Cout = cout; a.k = 1; a.stride = 1; a.pad = 0; a.act = silu ? 1 : 0;
    std::vector<ConvLaunch> cands;
    KCHK(plan_conv_candidates(a, &cands));
    const ConvLaunch& l = cands[(size_t)(plan_index < 0 ? 0 : plan_index) % cands.size()];
    if (n_plans) *n_plans = (int)cands.size();
    KCHK(run_conv(l, nullptr));
    HIPCHK(hipDeviceSynchronize());
    if (!half) {
        HIPCHK(hipMemcpy(yout.data(), d_y, yout.size() * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> hb(yout.size());
        HIPCHK(hipMemcpy(hb.data(), d_y, hb.size() * 2, hipMemcpyDeviceToHost));
        halfs_to_floats(hb.data(), yout.data(), hb.size());
    }
    for (size_t p = 0; p < np; ++p) std::memcpy(y + p * cout, &yout[p * cs_out], (size_t)cout * 4);
    return MI355_OK;
}

int mi355_op_conv1x1_upcat(int device_id, const float* x_half, const float* x_skip, int n, int h, int w, int up_c, int skip_c,
                           const float* w_oihw, const float* bias, int cout, int silu, float* y, int plan_index, int* n_plans) {
    return op_conv1x1_upcat_impl(device_id, x_half, x_skip, n, h, w, up_c, skip_c, w_oihw, bias, cout, silu, y, plan_index, n_plans, false);
}

int mi355_op_conv1x1_upcat_f16(int device_id, const float* x_half, const float* x_skip, int n, int h, int w, int up_c, int skip_c,
                               const float* w_oihw, const float* bias, int cout, int silu, float* y, int plan_index, int* n_plans) {
    return op_conv1x1_upcat_impl(device_id, x_half, x_skip, n, h, w, up_c, skip_c, w_oihw, bias, cout, silu, y, plan_index, n_plans, true);
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

static int op_stem_impl(int device_id, const uint8_t* bgr, int n, int h, int w, const float* w_oihw, const float* bias, int cout,
                        int k, int stride, bool half, int variant, void* y) {
    if (!bgr || !w_oihw || !bias || !y || n <= 0 || h <= 0 || w <= 0 || cout <= 0) return fail(MI355_EINVAL, "bad argument");
    if ((k != 3 && k != 6) || (h % stride) || (w % stride)) return fail(MI355_EINVAL, "k/stride not supported");
    HIPCHK(hipSetDevice(device_id));
    const int ho = h / stride, wo = w / stride, cs = round_up(cout, 4);
    const size_t es = half ? 2 : 4;
    DevMem dm; uint8_t* d_img; float *d_y, *d_w, *d_b, *d_l;
    const size_t ib = (size_t)n * h * w * 3, yn = (size_t)n * ho * wo * cs;
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (float)i / 255.0f;
    HIPCHK(dm.alloc(&d_img, ib)); HIPCHK(dm.alloc(&d_y, yn * es)); HIPCHK(dm.alloc(&d_w, (size_t)cout * 3 * k * k * 4));
    const int bn = round_up(cout, 16);                               // the engine's bias arrays are padded the same way
    HIPCHK(dm.alloc(&d_b, (size_t)bn * 4)); HIPCHK(dm.alloc(&d_l, sizeof(lut)));
    HIPCHK(hipMemcpy(d_img, bgr, ib, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_w, w_oihw, (size_t)cout * 3 * k * k * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_b, 0, (size_t)bn * 4));
    HIPCHK(hipMemcpy(d_b, bias, (size_t)cout * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_l, lut, sizeof(lut), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_y, 0, yn * es));
    StemArgs s{};
    if (half && k == 3) {
        std::vector<uint16_t> fr;
        stem3_weight_frags(w_oihw, cout, fr);
        uint16_t* d_f;
        HIPCHK(dm.alloc(&d_f, fr.size() * 2));
        HIPCHK(hipMemcpy(d_f, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
        s.wfrag = d_f;
    } else if (k == 3) {
        std::vector<float> fr;
        stem3_weight_frags_f32(w_oihw, cout, fr);
        float* d_f;
        HIPCHK(dm.alloc(&d_f, fr.size() * 4));
        HIPCHK(hipMemcpy(d_f, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
        s.wfrag = d_f;
    }
    s.img = d_img; s.dst = d_y; s.dst_cs = cs; s.w = d_w; s.bias = d_b; s.lut = d_l;
    s.B = n; s.H = h; s.W = w; s.Hout = ho; s.Wout = wo; s.Cout = cout; s.k = k; s.stride = stride; s.pad = (k == 6 ? 2 : k / 2);
    s.out_half = half ? 1 : 0; s.variant = variant;
    KCHK(launch_stem(s, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint8_t> yo(yn * es);
    HIPCHK(hipMemcpy(yo.data(), d_y, yn * es, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < (size_t)n * ho * wo; ++p) std::memcpy((uint8_t*)y + p * cout * es, &yo[p * cs * es], (size_t)cout * es);
    return MI355_OK;
}

int mi355_op_stem(int device_id, const uint8_t* bgr, int n, int h, int w, const float* w_oihw, const float* bias, int cout,
                  int k, int stride, float* y) {
    return op_stem_impl(device_id, bgr, n, h, w, w_oihw, bias, cout, k, stride, false, 0, y);
}

int mi355_op_stem_f16(int device_id, const uint8_t* bgr, int n, int h, int w, const float* w_oihw, const float* bias, int cout,
                      int k, int stride, int variant, uint16_t* y) {
    if (variant < 0 || (variant & 255) > 2 || (variant >> 8) > 31) return fail(MI355_EINVAL, "variant must be 0 (the launcher's choice), 1 (general) or 2 (k3 s2)");
    if ((variant & 255) == 2 && (k != 3 || stride != 2 || (w & 3))) return fail(MI355_EINVAL, "variant 2 is the k 3, stride 2 kernel for widths that are multiples of 4");
    return op_stem_impl(device_id, bgr, n, h, w, w_oihw, bias, cout, k, stride, true, variant, y);
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

