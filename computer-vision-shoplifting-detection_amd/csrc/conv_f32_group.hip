// Grouped launch: several independent convs (one depth of the op DAG) as ONE grid, each member on its own tuned kernel
// instance.  Device code of the members: conv_f32.h (the kernel bodies take virtual block coordinates).  See common.h.
#include "conv_f32.h"
#include "conv_f32_inst.h"
#include <algorithm>
#include <cstddef>

namespace mi355 {

// The menu: instances the autotuner picks for latency-bound launches (1 or 2 pixel tiles per wave).
//   kind = family base + index.  igemm 3x3: ((stride - 1) * 2 + f2) * 12 + (PT - 1) * 6 + (CT - 1) * 3 + log2(WP)
enum { G_IGEMM = 0, G_SPLITK = 48, G_STREAM = 56, G_KINDS = 65 };

__device__ __forceinline__ int log2i(int v) { return v == 4 ? 2 : v == 2 ? 1 : 0; }

#define MI355_G_IGEMM_CASES(st, f2, base)                                                                                   \
    case base + 0:  conv_igemm_f32_body<3, st, 1, 1, 1, f2>(a, lds, bid); break;                                             \
    case base + 1:  conv_igemm_f32_body<3, st, 1, 1, 2, f2>(a, lds, bid); break;                                             \
    case base + 2:  conv_igemm_f32_body<3, st, 1, 1, 4, f2>(a, lds, bid); break;                                             \
    case base + 3:  conv_igemm_f32_body<3, st, 1, 2, 1, f2>(a, lds, bid); break;                                             \
    case base + 4:  conv_igemm_f32_body<3, st, 1, 2, 2, f2>(a, lds, bid); break;                                             \
    case base + 5:  conv_igemm_f32_body<3, st, 1, 2, 4, f2>(a, lds, bid); break;                                             \
    case base + 6:  conv_igemm_f32_body<3, st, 2, 1, 1, f2>(a, lds, bid); break;                                             \
    case base + 7:  conv_igemm_f32_body<3, st, 2, 1, 2, f2>(a, lds, bid); break;                                             \
    case base + 8:  conv_igemm_f32_body<3, st, 2, 1, 4, f2>(a, lds, bid); break;                                             \
    case base + 9:  conv_igemm_f32_body<3, st, 2, 2, 1, f2>(a, lds, bid); break;                                             \
    case base + 10: conv_igemm_f32_body<3, st, 2, 2, 2, f2>(a, lds, bid); break;                                             \
    case base + 11: conv_igemm_f32_body<3, st, 2, 2, 4, f2>(a, lds, bid); break;

// One member's work: its kernel instance (`kind`) on its arguments.  Force-inlined once per member SLOT, so that `a` is always a
// by-value kernel parameter at a static offset of the kernarg segment: every field then arrives by on-demand scalar loads, exactly
// as in the stand-alone kernels.  (A first version copied the selected member's arguments out of an array: 75 dwords loaded up
// front, 370 SGPRs spilled to VGPR lanes -- and sporadically wrong accumulator lanes in the LDS-free streaming member.)
__device__ __forceinline__ void run_member(const ConvKArgs& a, int kind, float* lds, const BlockId& bid) {
    switch (kind) {
        MI355_G_IGEMM_CASES(1, false, 0)
        MI355_G_IGEMM_CASES(1, true, 12)
        MI355_G_IGEMM_CASES(2, false, 24)
        MI355_G_IGEMM_CASES(2, true, 36)
        case G_SPLITK + 0: conv_splitk_f32_body<3, 1, 1, 1>(a, lds, bid); break;
        case G_SPLITK + 1: conv_splitk_f32_body<3, 1, 1, 2>(a, lds, bid); break;
        case G_SPLITK + 2: conv_splitk_f32_body<3, 1, 2, 1>(a, lds, bid); break;
        case G_SPLITK + 3: conv_splitk_f32_body<3, 1, 2, 2>(a, lds, bid); break;
        case G_SPLITK + 4: conv_splitk_f32_body<3, 2, 1, 1>(a, lds, bid); break;
        case G_SPLITK + 5: conv_splitk_f32_body<3, 2, 1, 2>(a, lds, bid); break;
        case G_SPLITK + 6: conv_splitk_f32_body<3, 2, 2, 1>(a, lds, bid); break;
        case G_SPLITK + 7: conv_splitk_f32_body<3, 2, 2, 2>(a, lds, bid); break;
        case G_STREAM + 0: conv1x1_stream_f32_body<1, 1>(a, bid); break;      // register tiles of at most 4 MFMA tiles: the widest
        case G_STREAM + 1: conv1x1_stream_f32_body<1, 2>(a, bid); break;      // member sets the register count (= occupancy) of all
        case G_STREAM + 2: conv1x1_stream_f32_body<1, 4>(a, bid); break;
        case G_STREAM + 3: conv1x1_stream_f32_body<2, 1>(a, bid); break;
        case G_STREAM + 4: conv1x1_stream_f32_body<2, 2>(a, bid); break;
        case G_STREAM + 6: conv1x1_stream_f32_body<4, 1>(a, bid); break;
        default: break;
    }
}

#ifndef MI355_GROUP_MINWAVES
#define MI355_GROUP_MINWAVES 4      // the widest menu instance sets every member's register count: 4 waves per SIMD = 128 registers
#endif
__global__ __launch_bounds__(256, MI355_GROUP_MINWAVES) void conv_group_f32(GroupHdr hdr, ConvKArgs a0, ConvKArgs a1, ConvKArgs a2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(kGroupMax == 3, "one by-value argument block per member slot");
    // member of this block: block-uniform (scalar compares on blockIdx.x)
    const int m = (hdr.n > 2 && blockIdx.x >= hdr.base[2]) ? 2 : (hdr.n > 1 && blockIdx.x >= hdr.base[1]) ? 1 : 0;
    const unsigned local = blockIdx.x - hdr.base[m];
    const unsigned gx = hdr.gx[m], gy = hdr.gy[m];
    if (local >= gx * gy) return;                        // padding blocks (member ranges are rounded up to multiples of 8)
    const BlockId bid{local % gx, local / gx, gx, gy};
    const int kind = hdr.kind[m];
    if (m == 0) run_member(a0, kind, lds, bid);
    else if (m == 1) run_member(a1, kind, lds, bid);
    else run_member(a2, kind, lds, bid);
}

int group_kind(const ConvLaunch& l, int ks, int stride) {
    const bool f2 = l.a.w2 != nullptr;
    auto lg = [](int v) { return v == 4 ? 2 : v == 2 ? 1 : v == 1 ? 0 : -1; };
    if (l.version == 1 && ks == 3 && (l.PT == 1 || l.PT == 2) && (l.CT == 1 || l.CT == 2) && lg(l.WP) >= 0 && (stride == 1 || stride == 2))
        return G_IGEMM + ((stride - 1) * 2 + (f2 ? 1 : 0)) * 12 + (l.PT - 1) * 6 + (l.CT - 1) * 3 + lg(l.WP);
    if (l.version == 6 && ks == 3 && (l.PT == 1 || l.PT == 2) && (l.CT == 1 || l.CT == 2) && (stride == 1 || stride == 2))
        return G_SPLITK + (stride - 1) * 4 + (l.PT - 1) * 2 + (l.CT - 1);
    if (l.version == 3 && ks == 1 && lg(l.PT) >= 0 && lg(l.CT) >= 0 && l.PT * l.CT <= 4)
        return G_STREAM + lg(l.PT) * 3 + lg(l.CT);
    return -1;
}

const char* plan_group(const std::vector<ConvLaunch>& members, const std::vector<int>& kinds, GroupLaunch* out) {
    if (members.size() < 2 || members.size() > (size_t)kGroupMax || kinds.size() != members.size()) return "group: 2..3 members";
    GroupLaunch g{};
    g.n_members = g.k.hdr.n = (int)members.size();
    unsigned at = 0;
    size_t lds = 0;
    for (size_t m = 0; m < members.size(); ++m) {
        const ConvLaunch& l = members[m];
        if (kinds[m] < 0 || kinds[m] >= G_KINDS || l.threads != 256) return "group: member kernel is not on the menu";
        g.k.hdr.base[m] = at;
        g.k.hdr.kind[m] = kinds[m];
        g.k.hdr.gx[m] = l.grid_x; g.k.hdr.gy[m] = l.grid_y;
        g.k.a[m] = l.a;
        at += (l.grid_x * l.grid_y + 7u) & ~7u;
        lds = std::max(lds, l.lds);
    }
    for (size_t m = members.size(); m <= (size_t)kGroupMax; ++m) g.k.hdr.base[m] = at;
    for (size_t m = members.size(); m < (size_t)kGroupMax; ++m) g.k.a[m] = members[0].a;        // unused slots: any valid block
    if (at >= (1u << 24)) return "group: too many blocks";
    g.grid = at; g.lds = lds;
    *out = g;
    return nullptr;
}

const char* run_group(const GroupLaunch& g, hipStream_t st) {
    hipLaunchKernelGGL(conv_group_f32, dim3(g.grid), dim3(256), g.lds, st, g.k.hdr, g.k.a[0], g.k.a[1], g.k.a[2]);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace mi355
