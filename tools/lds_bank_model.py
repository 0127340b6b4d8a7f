"""ds_read_b128 bank-conflict model of the MI355X guide (four 16-lane groups, 64 four-byte banks): LDS cycles of a B-operand fragment read
(lane (p, g): 16 bytes of pixel p, channel group g) for padded pixel strides and for XOR-swizzled 64-byte records.  4 = conflict-free.
Behind csrc/conv_f16_lw.hip's halo layout and the MI355_LDS_PAD default; confirmed by SQ_LDS_BANK_CONFLICT."""
import itertools
G0 = list(range(0,4))+list(range(12,16))+list(range(20,28))
G1 = list(range(4,12))+list(range(16,20))+list(range(28,32))
GROUPS = [G0, G1, [l+32 for l in G0], [l+32 for l in G1]]
def cycles(addr_of_lane):
    tot = 0
    for grp in GROUPS:
        banks = {}
        for l in grp:
            a = addr_of_lane(l)
            b = (a//4) % 64
            banks.setdefault(b//4*4, set()).add(a)   # 16B aligned -> 4-bank unit
        tot += max(len(v) for v in banks.values())
    return tot   # 4 = conflict free
def test(stride, sfun, bases=range(0,64)):
    worst = 0; sumc = 0
    for pb in bases:
        def addr(l):
            p = l & 15; g = l >> 4; pix = pb + p
            return pix*stride + ((g ^ sfun(pix)) * 16)
        c = cycles(addr); worst = max(worst, c); sumc += c
    return worst, sumc/len(list(bases))
print("pad80 no swizzle", test(80, lambda pix: 0))
print("pad72", test(72, lambda pix: 0) if 72%16==0 else None)
for stride in (64, 80, 96, 112, 128, 144, 160, 192, 208):
    print(stride, test(stride, lambda pix: 0))
# swizzles on stride 64: s = table[(pix>>2)&3]
best = []
for tab in itertools.product(range(4), repeat=4):
    w, a = test(64, lambda pix: tab[(pix>>2)&3])
    best.append((w, a, tab))
best.sort(); print(best[:6])
best = []
for tab in itertools.product(range(4), repeat=8):
    w, a = test(64, lambda pix: tab[(pix>>1)&7])
    best.append((w, a, tab))
best.sort(); print(best[:4])
