// Grouped launch: several independent convs (one depth of the op DAG) as ONE grid, each member on its own tuned kernel
// instance.  Device code of the members: conv_f32.h (the kernel bodies take virtual block coordinates).  See common.h.
#include "conv_f32.h"
#include "conv_f32_inst.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>

namespace mi355 {

// The menu: instances the autotuner picks for latency-bound launches (1 or 2 pixel tiles per wave).
//   kind = family base + index.  igemm 3x3: ((stride - 1) * 2 + f2) * 12 + (PT - 1) * 6 + (CT - 1) * 3 + log2(WP)
enum { G_IGEMM = 0, G_SPLITK = 48, G_STREAM = 56, G_KINDS = 65 };

__device__ __forceinline__ int log2i(int v) { return v == 4 ? 2 : v == 2 ? 1 : 0; }

typedef const __attribute__((address_space(4))) ConvKArgs KArgsC;     // an argument block in the kernarg segment (constant address space)

#define MI355_G_IGEMM_CASES(st, f2, base)                                                                                   \
    case base + 0:  conv_igemm_f32_body<3, st, 1, 1, 1, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 1:  conv_igemm_f32_body<3, st, 1, 1, 2, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 2:  conv_igemm_f32_body<3, st, 1, 1, 4, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 3:  conv_igemm_f32_body<3, st, 1, 2, 1, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 4:  conv_igemm_f32_body<3, st, 1, 2, 2, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 5:  conv_igemm_f32_body<3, st, 1, 2, 4, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 6:  conv_igemm_f32_body<3, st, 2, 1, 1, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 7:  conv_igemm_f32_body<3, st, 2, 1, 2, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 8:  conv_igemm_f32_body<3, st, 2, 1, 4, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 9:  conv_igemm_f32_body<3, st, 2, 2, 1, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 10: conv_igemm_f32_body<3, st, 2, 2, 2, f2, KArgsC>(a, lds, bid); break;                                     \
    case base + 11: conv_igemm_f32_body<3, st, 2, 2, 4, f2, KArgsC>(a, lds, bid); break;

#ifndef MI355_GROUP_MINWAVES
#define MI355_GROUP_MINWAVES 4      // the widest menu instance sets every member's register count: 4 waves per SIMD = 128 registers
#endif
// The members' argument blocks travel as by-value kernel parameters; a block reads ITS member's block straight from the kernarg
// segment through a constant-address-space reference, field by field and on demand -- scalar loads at (uniform base + static
// offset), the way a stand-alone kernel reads its own arguments.  Two earlier forms loaded a whole block (or all three) up
// front: 370 - 1600 SGPR spills to VGPR lanes around the matrix instructions, and sporadically wrong accumulator lanes in the
// streaming member whenever a split-K member shared the launch (16 floats in ~10 % of the passes; tools/dbg_stress.py).
__global__ __launch_bounds__(256, MI355_GROUP_MINWAVES) void conv_group_f32(GroupKArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef const __attribute__((address_space(4))) char* KPtr;
    const KPtr kbase = (KPtr)__builtin_amdgcn_kernarg_segment_ptr();
    typedef const __attribute__((address_space(4))) GroupHdr HdrC;
    HdrC& hdr = *(HdrC*)(kbase + offsetof(GroupKArgs, hdr));
    // member of this block: block-uniform (scalar compares on blockIdx.x)
    int m = 0;
#pragma unroll
    for (int i = 1; i < kGroupMax; ++i)
        if (i < hdr.n && blockIdx.x >= hdr.base[i]) m = i;
    const unsigned local = blockIdx.x - hdr.base[m];
    const unsigned gx = hdr.gx[m], gy = hdr.gy[m];
    if (local >= gx * gy) return;                        // padding blocks (member ranges are rounded up to multiples of 8)
    const BlockId bid{local % gx, local / gx, gx, gy};
    const int kind = hdr.kind[m];
    KArgsC& a = *(KArgsC*)(kbase + offsetof(GroupKArgs, a) + (size_t)m * sizeof(ConvKArgs));
    (void)g;
    switch (kind) {
        MI355_G_IGEMM_CASES(1, false, 0)
        MI355_G_IGEMM_CASES(1, true, 12)
        MI355_G_IGEMM_CASES(2, false, 24)
        MI355_G_IGEMM_CASES(2, true, 36)
        case G_SPLITK + 0: conv_splitk_f32_body<3, 1, 1, 1, KArgsC>(a, lds, bid); break;
        case G_SPLITK + 1: conv_splitk_f32_body<3, 1, 1, 2, KArgsC>(a, lds, bid); break;
        case G_SPLITK + 2: conv_splitk_f32_body<3, 1, 2, 1, KArgsC>(a, lds, bid); break;
        case G_SPLITK + 3: conv_splitk_f32_body<3, 1, 2, 2, KArgsC>(a, lds, bid); break;
        case G_SPLITK + 4: conv_splitk_f32_body<3, 2, 1, 1, KArgsC>(a, lds, bid); break;
        case G_SPLITK + 5: conv_splitk_f32_body<3, 2, 1, 2, KArgsC>(a, lds, bid); break;
        case G_SPLITK + 6: conv_splitk_f32_body<3, 2, 2, 1, KArgsC>(a, lds, bid); break;
        case G_SPLITK + 7: conv_splitk_f32_body<3, 2, 2, 2, KArgsC>(a, lds, bid); break;
        case G_STREAM + 0: conv1x1_stream_f32_body<1, 1, KArgsC>(a, bid); break;      // register tiles of at most 4 MFMA tiles: the widest
        case G_STREAM + 1: conv1x1_stream_f32_body<1, 2, KArgsC>(a, bid); break;      // member sets the register count (= occupancy) of all
        case G_STREAM + 2: conv1x1_stream_f32_body<1, 4, KArgsC>(a, bid); break;
        case G_STREAM + 3: conv1x1_stream_f32_body<2, 1, KArgsC>(a, bid); break;
        case G_STREAM + 4: conv1x1_stream_f32_body<2, 2, KArgsC>(a, bid); break;
        case G_STREAM + 6: conv1x1_stream_f32_body<4, 1, KArgsC>(a, bid); break;
        default: break;
    }
}

int group_kind(const ConvLaunch& l, int ks, int stride) {
    // MI355_GROUP_MENU: bit 0 = LDS-staged 3x3, bit 1 = split-K, bit 2 = streaming pointwise, bit 3 = fused pointwise stage.
    // The streaming pointwise instances were off the menu for most of round 3: beside a split-K member their output showed zeroed
    // words in pixel lanes 12-15, in 7-100 % of the passes depending on the shape.  Cause (found on the half=True kernels, which
    // showed it in isolation): the data registers of a 16-byte buffer store were rewritten one instruction too early -- a head's
    // final conv has no activation between a store and the next tile's bias add, and the compiler inserts no wait state for stores
    // that take their offset from an SGPR (common.h:buffer_store_b128 does now).  300 stress passes over five model / batch / size
    // combinations with the streaming family on the menu: bit-exact (tools/dbg_stress.py).
    static const int menu = getenv("MI355_GROUP_MENU") ? atoi(getenv("MI355_GROUP_MENU")) : 15;
    const bool f2 = l.a.w2 != nullptr;
    if ((l.version == 1 && !(menu & 1)) || (l.version == 6 && !(menu & 2)) || (l.version == 3 && !(menu & 4)) || (f2 && !(menu & 8))) return -1;
    auto lg = [](int v) { return v == 4 ? 2 : v == 2 ? 1 : v == 1 ? 0 : -1; };
    if (l.version == 1 && ks == 3 && (l.PT == 1 || l.PT == 2) && (l.CT == 1 || l.CT == 2) && lg(l.WP) >= 0 && (stride == 1 || stride == 2))
        return G_IGEMM + ((stride - 1) * 2 + (f2 ? 1 : 0)) * 12 + (l.PT - 1) * 6 + (l.CT - 1) * 3 + lg(l.WP);
    if (l.version == 6 && ks == 3 && (l.PT == 1 || l.PT == 2) && (l.CT == 1 || l.CT == 2) && (stride == 1 || stride == 2))
        return G_SPLITK + (stride - 1) * 4 + (l.PT - 1) * 2 + (l.CT - 1);
    if (l.version == 3 && ks == 1 && !l.a.up_c && lg(l.PT) >= 0 && lg(l.CT) >= 0 && l.PT * l.CT <= 4)
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
    hipLaunchKernelGGL(conv_group_f32, dim3(g.grid), dim3(256), g.lds, st, g.k);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

}  // namespace mi355
