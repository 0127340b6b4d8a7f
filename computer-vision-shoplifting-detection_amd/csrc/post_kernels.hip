// Head decode (DFL + dist2bbox + sigmoid + keypoint decode) and non_max_suppression + scale-back.
//
// Restates, in the reference's own fp32 operation order (FP contraction is switched OFF in this file):
//   ultralytics/nn/modules/head.py:Detect._inference / Pose.kpts_decode, block.py:DFL, utils/tal.py:dist2bbox
//   ultralytics/utils/nms.py:non_max_suppression (+ torchvision.ops.nms CPU kernel: stable descending sort,
//   suppress IoU > thr, IoU = inter / (area_i + area_j - inter))
//   ultralytics/utils/ops.py:scale_boxes / clip_boxes / scale_coords / clip_coords
// reached from /root/reference/model.py:38-40.
#include "common.h"
#include "../../include/mi355_yolo.h"
#include "detmath.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace mi355 {

// ---------------------------------------------------------------------------------------------- decode
// Reads the raw head maps (NHWC, per level: 64 box logits | nc class logits | nk kpts) and writes, per anchor, the
// decoded row [4 box | nc scores | nk kpts] (anchor-major) plus (best score, best class).  FULL = also store the nc
// sigmoid scores (only the raw-head parity entry point needs them: NMS reads box, keypoints and `best` only), which cuts
// the kernel's HBM writes from 86 to 6 floats per anchor for an 80-class detector.
// FOUR lanes per anchor and no LDS.  Lane s of a quad owns DFL side s (its 16 logits are 64 contiguous bytes:
// four 16-byte loads), a quarter of the class logits and a quarter of the keypoint values; the four distances, the class
// maximum and the (score, first argmax) pair are combined with wave shuffles.  (Round 1's LDS-staged one-lane-per-anchor
// kernels staged 336 bytes per anchor and so ran at 1.5 waves per SIMD: 0.78 vs 0.36 ms per 256 frames; removed.)  A DFL
// side is evaluated sequentially by one lane in the canonical operation order.
template <bool FULL>
__global__ __launch_bounds__(256) void decode_kernel_quad(DecodeArgs a) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.A;
    const long ag0 = gid >> 2;
    const int s = (int)(gid & 3);
    const bool live = ag0 < total;
    const long ag = live ? ag0 : total - 1;                 // dead quads shadow the last anchor (shuffles need all lanes), stores are guarded
    const int b = (int)(ag / a.A), an = (int)(ag - (long)b * a.A);
    int l = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (j < a.n_levels && an >= a.lv[j].anchor0) l = j;
    const HeadLevelArgs lv = a.lv[l];
    const int li = an - lv.anchor0;
    const int y = li / lv.W, x = li - y * lv.W;
    const float ax = (float)x + 0.5f, ay = (float)y + 0.5f, st = (float)lv.stride;
    const float* src = lv.buf + ((size_t)b * lv.H * lv.W + li) * lv.cs;
    const int nk = a.nkpt * a.kdim, no = 4 + a.nc + nk;
    float* out = a.pred + (size_t)ag * no;
    const int lane = threadIdx.x & 63, qbase = lane & ~3;

    // ---- box: side s
    float d;
    {
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 q4 = *(const float4*)(src + lv.box_off + 16 * s + 4 * j);
            v[4 * j] = q4.x; v[4 * j + 1] = q4.y; v[4 * j + 2] = q4.z; v[4 * j + 3] = q4.w;
        }
        float m = v[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) m = fmaxf(m, v[j]);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = det_expf(v[j] - m); sum += v[j]; }
        d = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) d += (v[j] / sum) * (float)j;
    }
    const float d0 = __shfl(d, qbase), d1 = __shfl(d, qbase + 1), d2 = __shfl(d, qbase + 2), d3 = __shfl(d, qbase + 3);
    if (live && s == 0) {
        const float x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3;
        float4 o;
        o.x = ((x1 + x2) / 2.0f) * st;
        o.y = ((y1 + y2) / 2.0f) * st;
        o.z = (x2 - x1) * st;
        o.w = (y2 - y1) * st;
        if ((no & 3) == 0) *(float4*)out = o;
        else { out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w; }
    }
    // ---- classes: lane s owns classes c = s, s + 4, s + 8, ... (or whole float4 groups, below)
    {
        const float* cl = src + lv.cls_off;
        // 16-byte form when the class slice allows it: lane s owns the float4 groups s, s + 4, ... (classes 4q .. 4q+3)
        const bool vec = (a.nc & 3) == 0 && (lv.cls_off & 3) == 0;
        const int ncq = a.nc >> 2;
        float thr = -__builtin_huge_valf();
        // NMS only needs max_c sigmoid(logit_c) and its FIRST argmax.  sigmoid is monotone, so only classes whose logit is
        // within a hair of the largest one can hold or tie the maximum: evaluate the (35-instruction, bit-exact) sigmoid for
        // those alone -- 1-2 classes instead of 80.  The window is exact, not heuristic: up to 11 the fp32 sigmoid still
        // separates logits 0.01 apart by >= 2.8 ulp (det_expf is within 1.4 ulp of exp; tests/test_oracle_det.py checks the
        // property), beyond that it saturates towards 1.0f, so everything above 10.9 is a candidate; below -80 it
        // underflows and every class is one.
        if (!FULL) {
            float m = -__builtin_huge_valf();
            if (vec) {
                for (int q = s; q < ncq; q += 4) {
                    const float4 v = *(const float4*)(cl + 4 * q);
                    m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                }
            } else {
                for (int c = s; c < a.nc; c += 4) m = fmaxf(m, cl[c]);
            }
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            thr = m > 11.0f ? 10.9f : (m < -80.0f ? -__builtin_huge_valf() : m - 0.01f);
        }
        float best = -1.f; int bi = 0x7fffffff;
        if (vec) {
            for (int q = s; q < ncq; q += 4) {
                const float4 v = *(const float4*)(cl + 4 * q);
                float lg[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (FULL || lg[j] >= thr) {
                        lg[j] = det_sigmoid(lg[j]);
                        if (lg[j] > best) { best = lg[j]; bi = 4 * q + j; }
                    }
                }
                if (FULL && live) {
                    if ((no & 3) == 0) *(float4*)(out + 4 + 4 * q) = make_float4(lg[0], lg[1], lg[2], lg[3]);
                    else { out[4 + 4 * q] = lg[0]; out[5 + 4 * q] = lg[1]; out[6 + 4 * q] = lg[2]; out[7 + 4 * q] = lg[3]; }
                }
            }
        } else {
            for (int c = s; c < a.nc; c += 4) {
                const float lg = cl[c];
                if (FULL || lg >= thr) {
                    const float sc = det_sigmoid(lg);
                    if (FULL && live) out[4 + c] = sc;
                    if (sc > best) { best = sc; bi = c; }
                }
            }
        }
        // (max score, FIRST class that attains it) over the quad == the reference's ascending `>` scan
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
            const float ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (bi == 0x7fffffff) bi = 0;                        // no score beat -1 (all NaN): the reference leaves class 0
        if (live && s == 0) a.best[ag] = make_float2(best, (float)bi);
    }
    // ---- keypoints: element e = s, s + 4, ... of the nk values; e = k * kdim + component
    if (nk) {
        const float* kp = src + lv.kpt_off;
        for (int e = s; e < nk; e += 4) {
            const int k = e / a.kdim, comp = e - k * a.kdim;
            float v = kp[e];
            if (comp == 0) v = (v * 2.0f + (ax - 0.5f)) * st;
            else if (comp == 1) v = (v * 2.0f + (ay - 0.5f)) * st;
            else if (a.kdim == 3) v = det_sigmoid(v);
            if (live) out[4 + a.nc + e] = v;
        }
    }
}

const char* launch_decode(const DecodeArgs& a, bool full, hipStream_t st) {
    for (int l = 0; l < a.n_levels; ++l)
        if ((a.lv[l].cs & 3) || (a.lv[l].box_off & 3)) return "decode: box logits must be 16-byte aligned";
    const long lanes = (long)a.B * a.A * 4;
    const dim3 g((unsigned)((lanes + 255) / 256));
    if (full) hipLaunchKernelGGL(decode_kernel_quad<true>, g, dim3(256), 0, st, a);
    else      hipLaunchKernelGGL(decode_kernel_quad<false>, g, dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// [B][rows][cols] -> [B][cols][rows]  (Ultralytics [B, no, A] <-> anchor-major [B, A, no])
__global__ void transpose_kernel(const float* in, float* out, int B, int rows, int cols) {
    const long total = (long)B * rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const int r = (int)((i / cols) % rows);
        const int b = (int)(i / ((long)cols * rows));
        out[((size_t)b * cols + c) * rows + r] = in[i];
    }
}
const char* launch_transpose_pred(const float* in, float* out, int B, int rows, int cols, hipStream_t st) {
    const long total = (long)B * rows * cols;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(transpose_kernel, dim3(grid), dim3(256), 0, st, in, out, B, rows, cols);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

__global__ void best_kernel(const float* pred, int B, int A, int no, int nc, float2* best) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * A) return;
    const float* p = pred + (size_t)i * no + 4;
    float bs = -1.f; int bi = 0;
    for (int c = 0; c < nc; ++c) if (p[c] > bs) { bs = p[c]; bi = c; }
    best[i] = make_float2(bs, (float)bi);
}
const char* launch_best_from_pred(const float* pred, int B, int A, int no, int nc, float2* best, hipStream_t st) {
    const long total = (long)B * A;
    hipLaunchKernelGGL(best_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pred, B, A, no, nc, best);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------------------- NMS
// Kernel 1 (one 256-thread block per image): candidate filter (best score > conf, best class allowed),
// then a bitonic sort of the 64-bit keys (~score_bits << 32 | anchor): ascending key order == descending
// score with ties broken by ascending anchor index == torch's stable descending sort of the anchor-ordered
// candidate list.
__device__ void bitonic_sort(unsigned long long* v, int n2, int tid, int nthreads) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = v[i], b = v[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { v[i] = b; v[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

constexpr int NMS_LDS_KEYS = 4096;
constexpr int NMS_SORT_THREADS = 1024;

// compare-exchange steps j = j_hi, j_hi/2, .. 1 of merge stage k on the LDS-resident chunk [base, base + NMS_LDS_KEYS) of a
// longer sequence (directions follow the GLOBAL index)
__device__ void bitonic_steps_lds(unsigned long long* sk, int base, int k, int j_hi, int tid) {
    for (int j = j_hi; j > 0; j >>= 1) {
        for (int l = tid; l < NMS_LDS_KEYS; l += NMS_SORT_THREADS) {
            const int lxj = l ^ j;
            if (lxj > l) {
                const unsigned long long a = sk[l], b = sk[lxj];
                const bool up = ((base + l) & k) == 0;
                if ((a > b) == up) { sk[l] = b; sk[lxj] = a; }
            }
        }
        __syncthreads();
    }
}

// More candidates than fit in LDS (conf ~0.001 validation-style calls, or 1280x1280 inputs): the classic hybrid -- every
// compare-exchange at distance < NMS_LDS_KEYS runs on LDS-resident chunks, only the few long-distance steps of the last
// merge stages touch global memory (3 global round trips for 8192 keys instead of 91).
__device__ void bitonic_sort_hybrid(unsigned long long* keys, unsigned long long* sk, int n2, int tid) {
    constexpr int C = NMS_LDS_KEYS;
    for (int base = 0; base < n2; base += C) {                       // stages k = 2 .. C: chunk-local
        for (int l = tid; l < C; l += NMS_SORT_THREADS) sk[l] = keys[base + l];
        __syncthreads();
        for (int k = 2; k <= C; k <<= 1) bitonic_steps_lds(sk, base, k, k >> 1, tid);
        for (int l = tid; l < C; l += NMS_SORT_THREADS) keys[base + l] = sk[l];
        __syncthreads();
    }
    for (int k = 2 * C; k <= n2; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) {                      // partners live in different chunks
            for (int i = tid; i < n2; i += NMS_SORT_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
        for (int base = 0; base < n2; base += C) {
            for (int l = tid; l < C; l += NMS_SORT_THREADS) sk[l] = keys[base + l];
            __syncthreads();
            bitonic_steps_lds(sk, base, k, C >> 1, tid);
            for (int l = tid; l < C; l += NMS_SORT_THREADS) keys[base + l] = sk[l];
            __syncthreads();
        }
    }
}

// candidate filter of one image: keys[0 .. n) = (~score bits, anchor index) of the anchors that pass conf / class mask, padded with
// ~0 up to the next power of two n2; returns n2 (block-uniform), *n_out = n
__device__ int nms_collect(const NmsArgs& a, int b, unsigned long long* keys, int* scount, int tid, int* n_out) {
    if (tid == 0) *scount = 0;
    __syncthreads();
    const float2* best = a.best + (size_t)b * a.A;
    for (int an = tid; an < a.A; an += NMS_SORT_THREADS) {
        const float2 bc = best[an];
        bool ok = bc.x > a.conf;
        if (ok && a.class_mask) { const int c = (int)bc.y; ok = (a.class_mask[c >> 5] >> (c & 31)) & 1u; }
        if (ok) {
            const int pos = atomicAdd(scount, 1);
            keys[pos] = ((unsigned long long)(~__float_as_uint(bc.x)) << 32) | (unsigned)an;
        }
    }
    __syncthreads();
    const int n = *scount;
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += NMS_SORT_THREADS) keys[i] = ~0ull;
    __syncthreads();
    *n_out = n;
    return n2;
}

__global__ __launch_bounds__(NMS_SORT_THREADS) void nms_sort_kernel(NmsArgs a, int* cand_counts) {
    __shared__ unsigned long long skeys[NMS_LDS_KEYS];
    __shared__ int scount;
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned long long* keys = a.keys + (size_t)b * a.Apow2;
    int n = 0;
    const int n2 = nms_collect(a, b, keys, &scount, tid, &n);
    if (n2 > 1) {
        if (n2 <= NMS_LDS_KEYS) {
            for (int i = tid; i < n2; i += NMS_SORT_THREADS) skeys[i] = keys[i];
            __syncthreads();
            bitonic_sort(skeys, n2, tid, NMS_SORT_THREADS);
            for (int i = tid; i < n; i += NMS_SORT_THREADS) keys[i] = skeys[i];
        } else {
            bitonic_sort_hybrid(keys, skeys, n2, tid);
        }
    }
    if (tid == 0) cand_counts[b] = n < a.max_nms ? n : a.max_nms;     // "if n > max_nms: keep the top max_nms by conf"
}

// ---- the same sort spread over many blocks, for maps with more than 16384 anchors (1280x1280: 33,600 anchors, tens of thousands of
// candidates at low conf): one block per image sorting 32k keys took 0.68 ms per step, at any batch.  The bitonic network is cut into
// launches: collect (one block per image), chunk-local stages k <= 4096 (one block per 4096-key chunk, in LDS), then per merge stage
// k the few long-distance steps (j >= 4096, global memory, all chunks in parallel) and the LDS-resident rest (j < 4096).  Blocks
// beyond an image's padded length exit at once.  Keys are unique, so the result is the sequence the one-block sort gives.
__global__ __launch_bounds__(NMS_SORT_THREADS) void nms_collect_kernel(NmsArgs a, int* cand_counts, int* sort_len) {
    __shared__ int scount;
    const int b = blockIdx.x, tid = threadIdx.x;
    int n = 0;
    const int n2 = nms_collect(a, b, a.keys + (size_t)b * a.Apow2, &scount, tid, &n);
    if (tid == 0) { cand_counts[b] = n < a.max_nms ? n : a.max_nms; sort_len[b] = n2; }
}

__global__ __launch_bounds__(NMS_SORT_THREADS) void nms_chunk_sort_kernel(NmsArgs a, const int* sort_len, int chunks) {
    __shared__ unsigned long long sk[NMS_LDS_KEYS];
    const int b = blockIdx.x / chunks, base = (blockIdx.x % chunks) * NMS_LDS_KEYS, tid = threadIdx.x;
    const int n2 = sort_len[b];
    if (base >= n2 || n2 < 2) return;
    unsigned long long* keys = a.keys + (size_t)b * a.Apow2;
    const int len = n2 < NMS_LDS_KEYS ? n2 : NMS_LDS_KEYS;            // a short sequence is one (partial) chunk
    for (int l = tid; l < len; l += NMS_SORT_THREADS) sk[l] = keys[base + l];
    __syncthreads();
    for (int k = 2; k <= len; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int l = tid; l < len; l += NMS_SORT_THREADS) {
                const int lxj = l ^ j;
                if (lxj > l) {
                    const unsigned long long x = sk[l], y = sk[lxj];
                    const bool up = ((base + l) & k) == 0;
                    if ((x > y) == up) { sk[l] = y; sk[lxj] = x; }
                }
            }
            __syncthreads();
        }
    for (int l = tid; l < len; l += NMS_SORT_THREADS) keys[base + l] = sk[l];
}

__global__ __launch_bounds__(NMS_SORT_THREADS) void nms_global_step_kernel(NmsArgs a, const int* sort_len, int blocks_per_image, int k, int j) {
    const int b = blockIdx.x / blocks_per_image;
    const int n2 = sort_len[b];
    if (k > n2) return;
    unsigned long long* keys = a.keys + (size_t)b * a.Apow2;
    const int i = (blockIdx.x % blocks_per_image) * NMS_SORT_THREADS + threadIdx.x;       // one thread per lower partner
    // enumerate the indices whose bit j is clear: insert a 0 at bit position log2(j)
    const int lo = i & (j - 1), idx = ((i - lo) << 1) | lo;
    if (idx >= n2) return;
    const unsigned long long x = keys[idx], y = keys[idx | j];
    const bool up = (idx & k) == 0;
    if ((x > y) == up) { keys[idx] = y; keys[idx | j] = x; }
}

__global__ __launch_bounds__(NMS_SORT_THREADS) void nms_lds_steps_kernel(NmsArgs a, const int* sort_len, int chunks, int k) {
    __shared__ unsigned long long sk[NMS_LDS_KEYS];
    const int b = blockIdx.x / chunks, base = (blockIdx.x % chunks) * NMS_LDS_KEYS, tid = threadIdx.x;
    const int n2 = sort_len[b];
    if (k > n2 || base >= n2) return;
    unsigned long long* keys = a.keys + (size_t)b * a.Apow2;
    for (int l = tid; l < NMS_LDS_KEYS; l += NMS_SORT_THREADS) sk[l] = keys[base + l];
    __syncthreads();
    bitonic_steps_lds(sk, base, k, NMS_LDS_KEYS >> 1, tid);
    for (int l = tid; l < NMS_LDS_KEYS; l += NMS_SORT_THREADS) keys[base + l] = sk[l];
}

// Kernel 2 (one block of NMS_WAVES waves per image): greedy suppression over the sorted candidates in chunks of 64, stopping
// at max_det kept boxes, then scale-back and row packing.  The test of a chunk against the boxes kept so far -- the bulk of
// the work: candidates x kept IoUs, each with an IEEE division -- is split over the waves (wave w takes kept boxes w,
// w + NMS_WAVES, ...; the survivor masks are AND-ed through LDS); the greedy pass inside the chunk is inherently serial and is
// run redundantly by every wave, which keeps the kept count uniform without a broadcast.  Decisions are those of the
// sequential algorithm (torchvision.ops.nms order): a candidate dies iff some earlier kept box overlaps it.
constexpr int NMS_MAX_DET = 1024;
constexpr int NMS_WAVES = 8;

struct CandBox { float x1, y1, x2, y2, area; };

__device__ __forceinline__ bool iou_gt(const CandBox& i, const CandBox& j, float thr) {
    const float xx1 = fmaxf(i.x1, j.x1), yy1 = fmaxf(i.y1, j.y1);
    const float xx2 = fminf(i.x2, j.x2), yy2 = fminf(i.y2, j.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (i.area + j.area - inter);
    return ovr > thr;
}

// WAVES = waves of the block running it (NMS_WAVES stand-alone; the sort kernel's 16 when it runs behind the sort in one launch)
template <int WAVES>
__device__ __forceinline__ void nms_greedy_body(const NmsArgs& a, const int b, const int n) {
    constexpr int NMS_WAVES = WAVES;
    __shared__ float kx1[NMS_MAX_DET], ky1[NMS_MAX_DET], kx2[NMS_MAX_DET], ky2[NMS_MAX_DET], kar[NMS_MAX_DET];
    __shared__ int kan[NMS_MAX_DET];
    __shared__ unsigned long long survive[2][NMS_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long* keys = a.keys + (size_t)b * a.Apow2;
    const float* pred = a.pred + (size_t)b * a.A * a.no;
    const float2* best = a.best + (size_t)b * a.A;
    int nk = 0, it = 0;
    for (int base = 0; base < n && nk < a.max_det; base += 64, ++it) {
        const int j = base + lane;
        const bool valid = j < n;
        int an = 0;
        CandBox me = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid) {
            an = (int)(keys[j] & 0xffffffffull);
            const float* p = pred + (size_t)an * a.no;
            // xywh2xyxy, then boxes = x[:, :4] + cls * max_wh
            const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;
            const float c = best[an].y * a.max_wh;
            me.x1 = (p[0] - hw) + c; me.y1 = (p[1] - hh) + c;
            me.x2 = (p[0] + hw) + c; me.y2 = (p[1] + hh) + c;
            me.area = (me.x2 - me.x1) * (me.y2 - me.y1);
        }
        bool alive = valid;
        for (int k = wave; k < nk; k += NMS_WAVES) {               // this wave's share of the kept boxes
            const CandBox kb = {kx1[k], ky1[k], kx2[k], ky2[k], kar[k]};
            if (alive && iou_gt(kb, me, a.iou)) alive = false;
        }
        const unsigned long long mine = __ballot(alive);
        if (lane == 0) survive[it & 1][wave] = mine;
        __syncthreads();
        unsigned long long all = ~0ull;
#pragma unroll
        for (int w = 0; w < NMS_WAVES; ++w) all &= survive[it & 1][w];
        alive = (all >> lane) & 1ull;
        unsigned long long mask = all;
        while (mask) {                                              // greedy pass inside the chunk (every wave, identically)
            const int jj = __ffsll((long long)mask) - 1;
            CandBox kb;
            kb.x1 = __shfl(me.x1, jj); kb.y1 = __shfl(me.y1, jj); kb.x2 = __shfl(me.x2, jj); kb.y2 = __shfl(me.y2, jj);
            kb.area = __shfl(me.area, jj);
            if (wave == 0 && lane == jj) { kx1[nk] = me.x1; ky1[nk] = me.y1; kx2[nk] = me.x2; ky2[nk] = me.y2; kar[nk] = me.area; kan[nk] = an; }
            ++nk;
            if (nk >= a.max_det) break;
            if (alive && lane > jj && iou_gt(kb, me, a.iou)) alive = false;
            mask = __ballot(alive && lane > jj);
        }
        __syncthreads();                                            // the kept arrays are complete before the next chunk reads them
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.out_counts[b] = nk;
        if (a.host_counts) a.host_counts[b] = nk;       // small synchronous calls: the count goes straight to pinned host memory too
    }
    // Rows, one 32-bit word per thread step: consecutive lanes write consecutive words of a row, so the stores coalesce whether the
    // destination is HBM or -- small synchronous calls, where a copy engine hand-over costs more than the kernels -- pinned host
    // memory (out_rows then points there).  Every word is computed with the expression order of A.6 (xywh2xyxy, pad, gain, clip).
    constexpr int RW = (int)(sizeof(mi355_det) / 4);
    static_assert(sizeof(mi355_det) == 4 * (7 + MI355_MAX_KPT_FLOATS), "row words: x1 y1 x2 y2 conf cls anchor kpt[]");
    unsigned* rows_w = (unsigned*)((mi355_det*)a.out_rows + (size_t)b * a.max_det);
    const int nkf = a.nk < MI355_MAX_KPT_FLOATS ? a.nk : MI355_MAX_KPT_FLOATS;
    const int kd = a.kdim > 0 ? a.kdim : 1;
    for (int idx = threadIdx.x; idx < nk * RW; idx += 64 * NMS_WAVES) {
        const int k = idx / RW, q = idx - k * RW;
        const int an = kan[k];
        const float* p = pred + (size_t)an * a.no;
        unsigned w = 0u;
        if (q < 4) {
            const bool isx = (q & 1) == 0;
            const float half_ext = (isx ? p[2] : p[3]) / 2.0f;
            const float c = isx ? p[0] : p[1];
            float v = q < 2 ? c - half_ext : c + half_ext;
            if (a.scale_back) {
                v -= isx ? a.pad_x : a.pad_y;
                v /= a.gain;
                v = fminf(fmaxf(v, 0.f), isx ? a.orig_w : a.orig_h);
            }
            w = __float_as_uint(v);
        } else if (q == 4) {
            w = __float_as_uint(best[an].x);
        } else if (q == 5) {
            w = (unsigned)(int)best[an].y;
        } else if (q == 6) {
            w = (unsigned)an;
        } else if (q - 7 < nkf) {
            const int j = q - 7;
            float v = p[4 + a.nc + j];
            if (a.scale_back && (j % kd) < 2) {
                const bool isx = (j % kd) == 0;
                v -= isx ? a.kpad_x : a.kpad_y;
                v /= a.gain;
                v = fminf(fmaxf(v, 0.f), isx ? a.orig_w : a.orig_h);
            }
            w = __float_as_uint(v);
        }
        rows_w[idx] = w;
    }
}

__global__ __launch_bounds__(64 * NMS_WAVES) void nms_greedy_kernel(NmsArgs a, const int* cand_counts) {
    nms_greedy_body<NMS_WAVES>(a, blockIdx.x, cand_counts[blockIdx.x]);
}

// Sort and greedy suppression of one image in ONE launch (small calls: a launch costs more than either kernel's work; the greedy
// pass then runs on the sort's 16 waves, its share loop split 16 ways).  Same code, same order of operations, same rows.
__global__ __launch_bounds__(NMS_SORT_THREADS) void nms_sort_greedy_kernel(NmsArgs a, int* cand_counts) {
    __shared__ unsigned long long skeys[NMS_LDS_KEYS];
    __shared__ int scount;
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned long long* keys = a.keys + (size_t)b * a.Apow2;
    int n = 0;
    const int n2 = nms_collect(a, b, keys, &scount, tid, &n);
    if (n2 > 1) {
        if (n2 <= NMS_LDS_KEYS) {
            for (int i = tid; i < n2; i += NMS_SORT_THREADS) skeys[i] = keys[i];
            __syncthreads();
            bitonic_sort(skeys, n2, tid, NMS_SORT_THREADS);
            for (int i = tid; i < n; i += NMS_SORT_THREADS) keys[i] = skeys[i];
        } else {
            bitonic_sort_hybrid(keys, skeys, n2, tid);
        }
    }
    const int nc = n < a.max_nms ? n : a.max_nms;
    if (tid == 0) cand_counts[b] = nc;
    __syncthreads();                                   // the block's sorted keys (global memory) are visible to all of its threads
    nms_greedy_body<NMS_SORT_THREADS / 64>(a, b, nc);
}

static bool nms_fused_on() { static const bool on = !(getenv("MI355_NMS_FUSED") && atoi(getenv("MI355_NMS_FUSED")) == 0); return on; }

const char* launch_nms(const NmsArgs& a, hipStream_t st) {
    if (a.max_det < 1 || a.max_det > NMS_MAX_DET) return "nms: max_det must be in [1, 1024]";
    // cand_counts lives in the tail of out_counts' allocation: out_counts[B .. 2B)
    int* cand_counts = a.out_counts + a.B;
    if (a.Apow2 > 4 * NMS_LDS_KEYS) {
        // big maps: the sort as a sequence of wide launches (see nms_collect_kernel); sort lengths live at out_counts[2B .. 3B)
        int* sort_len = a.out_counts + 2 * a.B;
        const int chunks = a.Apow2 / NMS_LDS_KEYS;
        hipLaunchKernelGGL(nms_collect_kernel, dim3(a.B), dim3(NMS_SORT_THREADS), 0, st, a, cand_counts, sort_len);
        hipLaunchKernelGGL(nms_chunk_sort_kernel, dim3(a.B * chunks), dim3(NMS_SORT_THREADS), 0, st, a, sort_len, chunks);
        const int bpi = a.Apow2 / 2 / NMS_SORT_THREADS;
        for (int k = 2 * NMS_LDS_KEYS; k <= a.Apow2; k <<= 1) {
            for (int j = k >> 1; j >= NMS_LDS_KEYS; j >>= 1)
                hipLaunchKernelGGL(nms_global_step_kernel, dim3(a.B * bpi), dim3(NMS_SORT_THREADS), 0, st, a, sort_len, bpi, k, j);
            hipLaunchKernelGGL(nms_lds_steps_kernel, dim3(a.B * chunks), dim3(NMS_SORT_THREADS), 0, st, a, sort_len, chunks, k);
        }
    } else if (a.B <= 16 && nms_fused_on()) {
        // small calls: sort + greedy pass in one launch (a dependent launch costs 2.9 us + the kernel's ramp: DESIGN.md 3.5)
        hipLaunchKernelGGL(nms_sort_greedy_kernel, dim3(a.B), dim3(NMS_SORT_THREADS), 0, st, a, cand_counts);
        hipError_t e1 = hipGetLastError();
        return e1 == hipSuccess ? nullptr : hipGetErrorString(e1);
    } else {
        hipLaunchKernelGGL(nms_sort_kernel, dim3(a.B), dim3(NMS_SORT_THREADS), 0, st, a, cand_counts);
    }
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(a.B), dim3(64 * NMS_WAVES), 0, st, a, cand_counts);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ---------------------------------------------------------------------------------------------- row compaction
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int* counts, int n, int* offsets) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    int sum = 0;
    for (int i = tid * per; i < n && i < (tid + 1) * per; ++i) sum += counts[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {              // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = tid * per; i < n && i < (tid + 1) * per; ++i) { offsets[i] = run; run += counts[i]; }
    if (tid == 1023) offsets[n] = part[1023];
}

__global__ __launch_bounds__(256) void compact_rows_kernel(const unsigned* rows, const int* counts, const int* offsets, int max_det,
                                                           int row_words, unsigned* packed) {
    const int i = blockIdx.x;
    const int words = counts[i] * row_words;
    const unsigned* src = rows + (size_t)i * max_det * row_words;
    unsigned* dst = packed + (size_t)offsets[i] * row_words;
    for (int w = threadIdx.x; w < words; w += 256) dst[w] = src[w];
}

const char* launch_compact_rows(const void* rows, const int* counts, int n, int max_det, int row_words, int* offsets, void* packed,
                                hipStream_t st) {
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, st, counts, n, offsets);
    hipLaunchKernelGGL(compact_rows_kernel, dim3(n), dim3(256), 0, st, (const unsigned*)rows, counts, offsets, max_det, row_words,
                       (unsigned*)packed);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace mi355
