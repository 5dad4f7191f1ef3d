"""LDS bank-conflict model for the MFMA fragment reads of this repo (CPU tool, no GPU needed).

Model (MI355X_MICROARCH.md, LDS table): ds_read_b128 of a wave64 is serviced in four groups of 16 lanes -
{0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63} - over 64 four-byte banks; each distinct address on a
busy bank inside a group costs one extra cycle.  A fragment read has lane l at (row l&15, 16-byte k-chunk l>>4).

    python tools/lds_sim.py            # pitch table for BK = 32 / 64 and the XOR swizzle an unpadded (LDS-DMA) tile would need
"""
import itertools

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def cycles_b128(addr_of_lane):
    """addr_of_lane(l) -> byte address (16-byte aligned).  Returns LDS cycles of one ds_read_b128 (4 = conflict-free)."""
    total = 0
    for g in GROUPS:
        per_bank = {}
        for l in g:
            a = addr_of_lane(l)
            for d in range(4):
                per_bank.setdefault(((a >> 2) + d) & 63, set()).add(a + 4 * d)
        total += max(len(v) for v in per_bank.values())
    return total


def fragment_read(pitch_halfs, k0_halfs=0, swizzle=None):
    def addr(l):
        row, chunk = l & 15, (l >> 4) + (k0_halfs >> 3)
        if swizzle:
            chunk = swizzle(row, chunk)
        return (row * pitch_halfs + chunk * 8) * 2
    return cycles_b128(addr)


def worst_over_ksteps(pitch_halfs, bk, swizzle=None):
    return max(fragment_read(pitch_halfs, k0, swizzle) for k0 in range(0, bk, 32))


if __name__ == '__main__':
    for bk in (32, 64):
        print(f'BK={bk}: pad -> cycles per fragment read (4 = conflict-free)')
        for pad in (0, 8, 16, 24, 32):
            print(f'   pad {pad:2d}: {worst_over_ksteps(bk + pad, bk)}')
    # unpadded tiles (what buffer_load ... lds writes: 1 KB contiguous per wave instruction) need a swizzle
    for bk in (32, 64):
        nchunk = bk // 8
        best = None
        for mul, shift in itertools.product(range(1, 8), range(0, 4)):
            sw = lambda r, c, mul=mul, shift=shift: (c ^ ((r * mul) >> shift)) % nchunk
            cyc = worst_over_ksteps(bk, bk, sw)
            if best is None or cyc < best[0]:
                best = (cyc, mul, shift)
        print(f'BK={bk} unpadded: best XOR swizzle chunk ^= (row*{best[1]})>>{best[2]} mod {nchunk}: {best[0]} cycles')
