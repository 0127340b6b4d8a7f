// Shared declarations of the engine's translation units (host side + kernel launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace mi355 {

// ---- conv (implicit GEMM on the fp32 MFMA pipe), conv_igemm.hip ------------------------------------------
// Activations are NHWC fp32; a tensor is a channel slice of a buffer: base pointer already offset to the
// slice's first channel, `cs` = pixel stride of the buffer in floats (multiple of 4).
struct ConvArgs {
    const float* src; int src_cs;
    float* dst;       int dst_cs;
    const float* res; int res_cs;      // residual slice added after the activation, or nullptr
    const float* wpk;                  // packed weights (pack_conv_weights)
    const float* bias;                 // padded to a multiple of 16 floats
    int B, Hin, Win, Hout, Wout;
    int Cin, Cout;
    int k, stride, pad, act;           // k in {1,3}; act: 0 none, 1 SiLU (canonical, bit-reproducible), 2 SiLU on v_exp / v_rcp (fp32 kernels, opts.fast_act)
    const float* zeros;                // device pointer to >= 16 zero bytes (required)
    // half=True path (conv_igemm_f16.hip): dtype 1 = src / res / wpk hold fp16 (pack_conv_weights_f16), strides count halfs;
    // dst holds fp16 too unless out_f32 (the head's final convs).  The pointers keep their float* type; only bytes matter.
    int dtype = 0, out_f32 = 0;
    // Fused nearest-2x upsample on the read side (pointwise convs, v4 kernels only): input channels [0, up_c) come from
    // `src2`, a map of half the resolution (pixel (y/2, x/2), stride src2_cs), channels >= up_c from `src` as usual --
    // torch.nn.Upsample + Concat + Conv1x1 of the neck without the upsampled tensor ever being written.
    const float* src2 = nullptr; int src2_cs = 0, up_c = 0;
    // A pointwise conv fused behind this one (3x3 convs): Cout2 channels from packed weights w2 / bias2 into dst2; this
    // conv's own dst is then not written (its only reader is that 1x1).  f2_cout = 0: no fusion.
    const float* f2_wpk = nullptr; const float* f2_bias = nullptr; float* f2_dst = nullptr; int f2_dst_cs = 0, f2_cout = 0, f2_act = 0;
    int f2_out_f32 = 0;                // half=True: the fused pointwise conv writes fp32 (head outputs)
    // fp32: the fused pointwise conv reads f2_lead_c MORE input channels in FRONT of this conv's output (C2f.cv2 over cat(ys):
    // the earlier slices of the concat buffer), from f2_lead (same resolution as this conv's output, pixel stride f2_lead_cs);
    // f2_lead_c is a multiple of 16 and f2_wpk is packed for f2_lead_c + Cout input channels.  0: the pointwise conv reads
    // exactly this conv's output.
    const float* f2_lead = nullptr; int f2_lead_cs = 0, f2_lead_c = 0;
};
// number of floats pack_conv_weights writes: ceil(cout/16) * k*k * ceil(cin/16) * 256
size_t packed_weight_floats(int cout, int cin, int k);
// OIHW fp32 -> MFMA fragment order [cout_tile][tap][cin_block][lane(64)][4]
void pack_conv_weights(const float* w_oihw, int cout, int cin, int k, float* out);
// fp16 variant: ceil(cout/16) * k*k * ceil(cin/32) * 512 halfs, [cout_tile][tap][cin_block32][lane(64)][8], RNE from fp32
size_t packed_weight_halfs(int cout, int cin, int k);
void pack_conv_weights_f16(const float* w_oihw, int cout, int cin, int k, uint16_t* out_bits);
// fp16 kernels, Cout % 32 == 0: the rows of each PAIR of 16-cout MFMA tiles are interleaved (tile 2j row 4g+i = cout
// 32j+8g+i, tile 2j+1 row 4g+i = cout 32j+8g+4+i), so that a lane's 4+4 outputs are 8 consecutive couts = one 16-byte fp16
// store.  A pure relabelling of weight rows done at pack time; the packer and the kernels both ask this function.
__host__ __device__ inline bool conv_f16_pairs(int cout) { return (cout % 32) == 0; }
void floats_to_halfs(const float* in, uint16_t* out_bits, size_t n);     // round-to-nearest-even
void halfs_to_floats(const uint16_t* in_bits, float* out, size_t n);
const void* pick_conv_kernel_f16(int ks, int stride, int CT, int WP, int version, int stream_pt);
const void* pick_conv_pipe_f16(int CT, int WP, bool single, bool nkk8);
const void* pick_conv_fused_f16(int stride, int CT, int WP, int PT);       // conv_f16_fused.hip; PT 0 (= 4) or 8
const void* pick_conv_small_f16(int ks, int stride, int CT, int WP, int PT);   // conv_f16_small.hip; PT 1 or 2, CT 1 or 2
const void* pick_conv1x1_lwx_f16(int CT, bool up);                                     // conv_f16_lw.hip (version 10: pointwise, shared weights through LDS, pixels staged in full lines through wave-private LDS); CT 3 or 6
const void* pick_conv_lw_f16(int CT);                                         // conv_f16_lw.hip (version 7: weights in LDS, persistent); CT 3, 4 or 6
// Division of a block-uniform number by a launch constant on the SCALAR unit (tile decomposition of the conv kernels; the
// compiler's own lowering of an integer division runs on the vector ALU, which the fp32 matrix instructions share):
// M = ceil(2^40 / d) as (ml, mh); exact for n < 2^24, 1 <= d < 2^16.
struct FastDiv { unsigned ml, mh; };
inline FastDiv make_fastdiv(unsigned d) {
    const unsigned long long M = ((1ull << 40) + d - 1) / d;
    return FastDiv{(unsigned)M, (unsigned)(M >> 32)};
}
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned fastdiv(unsigned n, FastDiv f) { return (__umulhi(n, f.ml) + n * f.mh) >> 8; }
#endif
struct ConvKArgs {
    const float* src; float* dst; const float* res; const float* wpk; const float* bias;
    int src_cs, dst_cs, res_cs;
    int Hin, Win, Hout, Wout, Cin, Cout;
    int cib, n_ctiles, cin4;
    int ck, ck4_shift, ldp;
    int TW, TH, tiles_x, tiles_y, TWin, npix_in;
    float inv_TW, inv_TWin;
    int pad, act;
    const float* zeros;        // >= 16 bytes of zeros: what out-of-image / beyond-Cin slots read (loads stay unconditional)
    int out_f32;               // fp16 kernels: destination (and residual) hold fp32
    const float* src2; int src2_cs, up_c, up_W, up_H;   // fused upsample-on-read (v4): full-resolution W, H of the conv input
    int lds_buf_floats;        // offset of the second LDS region in 4-byte units (split-K partials / fused-1x1 image); fp16 -DMI355_F16_DIAG, unfused: flags
    int cgroups;               // conv_igemm_f32: groups of CT cout tiles a wave walks over one staged input (>= 1)
    // conv_igemm_f32 / conv_igemm_f16 <..., F2 = true>: the pointwise conv fused behind this one (packed weights, bias, destination slice)
    const float* w2; const float* bias2; float* dst2; int dst2_cs, Cout2, n_ctiles2, cib2, act2, ldp2, out2_f32;
    const float* lead; int lead_cs, lead_cib;   // fused pointwise stage: its first lead_cib k-blocks come from this slice in global memory (cib2 counts them)
    int n_tiles_total;         // B * tiles_x * tiles_y (persistent kernels walk tiles blockIdx.x, + gridDim.x, ...)
    FastDiv fd_tx, fd_ty, fd_gy;   // conv_igemm_f32: scalar division by tiles_x, tiles_y, gridDim.y
    int img_src, img_dst, img_res; // conv_igemm_f32: elements per image of the source / destination / residual slices' buffers (H * W * cs)
    int THin, st_rpi, st_nseg; float inv_row_slots;   // conv_igemm_f16 staging: halo rows, rows / row segments per 256-thread pass, 1 / (16-byte slots per halo row)
};
// A planned launch: kernel instance, grid, LDS bytes and kernel arguments.  Planning (tile / wave-arrangement
// search) is done once per (op, shape) by the engine; run_conv only enqueues.
struct ConvLaunch { const void* fn; unsigned grid_x, grid_y; size_t lds; ConvKArgs a; int CT, WP; double flops; int threads; int version; int PT; };
// ---- grouped launches (conv_f32_group.hip) ---------------------------------------------------------------------------
// Independent convs of one depth of the op DAG (the box / class / keypoint branches of the head levels beside the neck's own
// chain) run as ONE grid: block ranges [base[m], base[m + 1]) belong to member m, which runs its own tuned kernel instance
// (`kind` = index into a fixed menu of conv_igemm_f32 / conv_splitk_f32 / conv1x1_stream_f32 instances) on its own arguments.
// In the latency-bound regime (a few frames per pass) a launch costs 5-15 us whatever it computes and leaves most CUs idle; a
// group costs max(members) instead of sum(members) -- what several streams would buy without their cross-queue event waits.
constexpr int kGroupMax = 3;
struct GroupHdr {
    int n;
    unsigned base[kGroupMax + 1];      // first block of member m (multiples of 8: the XCD-aware work order stays valid per member)
    int kind[kGroupMax];
    unsigned gx[kGroupMax], gy[kGroupMax];
};
struct GroupKArgs { GroupHdr hdr; ConvKArgs a[kGroupMax]; };       // each a[m] travels as a by-value kernel parameter of its own
struct GroupLaunch { GroupKArgs k; unsigned grid; size_t lds; int n_members; int op[kGroupMax]; };
// menu index of a planned launch, or -1 when its kernel instance is not part of the group kernel
int group_kind(const ConvLaunch& l, int ks, int stride);
const char* plan_group(const std::vector<ConvLaunch>& members, const std::vector<int>& kinds, GroupLaunch* out);
const char* run_group(const GroupLaunch& g, hipStream_t st);

const char* plan_conv(const ConvArgs& c, ConvLaunch* out);
const char* plan_conv_candidates(const ConvArgs& c, std::vector<ConvLaunch>* out);   // best static guess first
const char* run_conv(const ConvLaunch& l, hipStream_t st);

// ---- misc kernels, misc_kernels.hip ---------------------------------------------------------------------
struct StemArgs {
    const uint8_t* img;                // letterboxed BGR u8 [B][H][W][3], dense
    float* dst; int dst_cs;
    const float* w;                    // device, OIHW fused [Cout][3][k][k]
    const float* bias;                 // device [Cout]
    const float* lut;                  // device [256]: (float)i / 255.0f
    int B, H, W, Hout, Wout, Cout, k, stride, pad;
    int out_half = 0;                  // 1: dst holds fp16 (dst_cs counts halfs); arithmetic stays fp32, one rounding on the store
    int fast_act = 0;                  // fp32 output: 1 = SiLU on v_exp / v_rcp (opts.fast_act) instead of the canonical form
    int variant = 0;                   // half output: 0 = the launcher's choice, 1 = the general kernel, 2 = the k3 s2 kernel (tests)
    const void* wfrag = nullptr;       // device: stem3_weight_frags() (half output) / stem3_weight_frags_f32() of w (k 3 stems; without it the general kernel runs)
};
void stem3_weight_frags(const float* w_oihw, int cout, std::vector<uint16_t>& out);   // [ceil(cout/16)][64][8] fp16 bit patterns
void stem3_weight_frags_f32(const float* w_oihw, int cout, std::vector<float>& out);   // [ceil(cout/16)][7][64] floats (fp32 k3 stems)
const char* launch_stem(const StemArgs& a, hipStream_t st);
const char* launch_upsample2x(const float* src, int src_cs, float* dst, int dst_cs, int B, int H, int W, int C,
                              hipStream_t st);
// SPPF: x1 = maxpool5(x0), x2 = maxpool5(x1), x3 = maxpool5(x2) (stride 1, -inf padding); src = x0 slice (C ch),
// dst = slice of 3*C channels receiving x1|x2|x3.
const char* launch_sppf_pools(const float* src, int src_cs, float* dst, int dst_cs, int B, int H, int W, int C,
                              hipStream_t st);
// fp16 buffers (strides / C in halfs, C % 8 == 0); max is exact in any precision
const char* launch_sppf_pools_f16(const void* src, int src_cs, void* dst, int dst_cs, int B, int H, int W, int C,
                                  hipStream_t st);
struct LetterboxArgs {
    const uint8_t* src; int H, W; long long frame_stride; int row_stride;   // source frames
    uint8_t* dst; int Hd, Wd;                                               // letterboxed output (dense)
    int top, left, Hr, Wr;                                                  // resized region placement / size
    const int* xtab;   // device [Wr*3]: x0, coef0, coef1 (11-bit fixed point)
    const int* ytab;   // device [Hr*3]
    int resize;        // 0: plain copy of the source into the region
    int B;
};
const char* launch_letterbox(const LetterboxArgs& a, hipStream_t st);

// ---- head decode + NMS, post_kernels.hip ----------------------------------------------------------------
struct HeadLevelArgs { const float* buf; int cs; int box_off, cls_off, kpt_off; int H, W, stride, anchor0; };
struct DecodeArgs {
    HeadLevelArgs lv[4]; int n_levels;
    int B, A, nc, nkpt, kdim;
    float* pred;        // [B][A][no] anchor-major decoded tensor (xywh, scores, kpts), no = 4+nc+nkpt*kdim
    float2* best;       // [B][A] (best score, best class as float)
};
// full = also store the nc class scores into pred (raw-head entry point); NMS itself only needs box/kpts/best
const char* launch_decode(const DecodeArgs& a, bool full, hipStream_t st);
// pred in Ultralytics layout [B][no][A] -> anchor-major [B][A][no] plus best[]; used by mi355_op_nms
const char* launch_best_from_pred(const float* pred_anchor_major, int B, int A, int no, int nc, float2* best,
                                  hipStream_t st);
const char* launch_transpose_pred(const float* in, float* out, int B, int rows, int cols, hipStream_t st);
// rows [n][max_det] (first counts[i] valid) -> packed [sum counts] in frame order; offsets[n + 1] = exclusive scan of counts.
// The host then copies sum(counts) rows instead of n * max_det (35 MB -> ~3 MB per 512-frame step).
const char* launch_compact_rows(const void* rows, const int* counts, int n, int max_det, int row_words, int* offsets, void* packed,
                                hipStream_t st);

struct NmsArgs {
    const float* pred; const float2* best;     // as written by decode
    int B, A, no, nc, nk, kdim;
    float conf, iou; int max_det; int max_nms; float max_wh;
    const unsigned* class_mask;                // device bitmask over classes, or nullptr = all
    unsigned long long* keys;                  // scratch [B][Apow2]
    int Apow2;
    // scale-back (A.6); identity when scale_back == 0
    int scale_back; float gain; float pad_x, pad_y, kpad_x, kpad_y; float orig_w, orig_h;
    void* out_rows;                            // device mi355_det [B][max_det]
    int* out_counts;                           // device [B] (+ [B, 3B): scratch of the sort kernels)
    int* host_counts = nullptr;                // optional: pinned host [B], written by the greedy kernel beside out_counts
};
const char* launch_nms(const NmsArgs& a, hipStream_t st);

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

#if defined(__HIPCC__)
// 16-byte buffer store whose data registers may be rewritten right after it.  gfx950 (measured, tools/dbg_conv_plans.py): the next vector
// instruction may NOT overwrite the VGPRs a buffer_store_dwordx4 takes its data from, also when the store takes its offset from an SGPR
// -- the one form for which the compiler's hazard recogniser inserts no wait state (its rule for "VMEM store of more than 8 bytes, data
// registers overwritten" exempts stores with an SGPR offset).  A head's final conv (no activation: bias add of the next tile two
// instructions behind the store) then stored the NEXT tile's bits / zeros in dword 0 of pixel lanes 12-15, in 1-20 % of the launches
// of some plans; one wait state was enough in 8 of 8 runs, two are inserted.  Round 3's "zeroed lanes 12-15" of the streaming
// pointwise member of a grouped launch was this.
#ifndef MI355_STORE_WAIT
#define MI355_STORE_WAIT 1
#endif
typedef unsigned mi355_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buffer_store_b128(mi355_u32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
#if MI355_STORE_WAIT                       // -DMI355_STORE_WAIT=0 (tools/ab_build.sh): the regression tests then fail, which is how they are checked
    __builtin_amdgcn_sched_barrier(0);     // nothing that precedes the store in program order may be moved into the store .. s_nop window
#endif
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, 0);
#if MI355_STORE_WAIT
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}
#endif

#if defined(__HIPCC__)
// XCD-aware work order for the (tile, cout group) grids of the conv kernels.  Workgroups are dealt round-robin over the 8 XCDs
// in dispatch order (x fastest, then y), so the blocks that share one XCD's L2 are b, b + 8, ...  Each of those eight
// sequences is given a CONTIGUOUS run of the logical order "tile-major, cout group innermost": the groups of one input tile
// (which stage the same pixels) and then its x neighbour (which shares the halo columns) follow each other on the same L2.
// Bijective for any grid size; a pure speed choice -- nothing depends on where a block runs.
#ifndef MI355_XCD_REMAP
#define MI355_XCD_REMAP 1
#endif
// A kernel body may run as a MEMBER of a grouped launch (conv_f32_group.hip: independent convs of one DAG depth in one grid):
// its block coordinates are then virtual -- (x, y) inside a (gx, gy) grid of its own, `lin0` = the launch-wide linear id of
// the member's first block (a multiple of 8, so that "blocks b and b + 8 share an XCD" still holds inside the member).
struct BlockId { unsigned x, y, gx, gy; };
#define MI355_BLOCK_ID() (::mi355::BlockId{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y})
// the same with the division by gridDim.y on the scalar unit (fd = make_fastdiv(gridDim.y); total blocks < 2^24)
__device__ __forceinline__ void xcd_work_item(int& tile, int& cgroup, FastDiv fd, const BlockId& bid) {
#if MI355_XCD_REMAP
    const unsigned gx = bid.gx, gy = bid.gy, n = gx * gy;
    const unsigned id = bid.y * gx + bid.x;
    const unsigned q = n >> 3, r = n & 7, x = id & 7;
    const unsigned logical = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    tile = (int)fastdiv(logical, fd);
    cgroup = (int)(logical - (unsigned)tile * gy);
#else
    tile = (int)bid.x; cgroup = (int)bid.y;
#endif
}
__device__ __forceinline__ void xcd_work_item(int& tile, int& cgroup, const BlockId& bid) {
#if MI355_XCD_REMAP
    const unsigned gx = bid.gx, gy = bid.gy, n = gx * gy;
    const unsigned id = bid.y * gx + bid.x;
    const unsigned q = n >> 3, r = n & 7, x = id & 7;
    const unsigned logical = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    tile = (int)(logical / gy);
    cgroup = (int)(logical - (unsigned)tile * gy);
#else
    tile = (int)bid.x; cgroup = (int)bid.y;
#endif
}
__device__ __forceinline__ void xcd_work_item(int& tile, int& cgroup, FastDiv fd) { xcd_work_item(tile, cgroup, fd, MI355_BLOCK_ID()); }
__device__ __forceinline__ void xcd_work_item(int& tile, int& cgroup) { xcd_work_item(tile, cgroup, MI355_BLOCK_ID()); }
#endif

}  // namespace mi355
