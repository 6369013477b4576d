"""CPU: the LDS image of the second-generation attention kernels (prompt_tts_amd/csrc/attn2.h, struct A2) by enumeration.

One image serves the row reads (ds_read_b128 of a 32x32x16 A operand) AND the transposed reads (ds_read_b64_tr_b16) of the K / V /
Q / dO tiles.  The header claims, for D = 32, 64, 128: both kinds of read are bank-conflict free, the swizzle permutes chunks
inside a row (so the LDS-DMA fill is a bijection), and tiles 16 / 32 rows further differ from the per-lane offsets by an immediate.
The formulas below restate A2<D>::sw / off; the lane groups are the hardware's (MI355X_MICROARCH.md, LDS table)."""
import pytest


def sw(r, D):
    RB = 2 * D; RPL = 1 if RB >= 256 else 256 // RB; NB = RB // 64
    return (((r // RPL) & (NB - 1)) << 2) | ((r >> 2) & 3)


def off(r, c, D):
    RB = 2 * D; NCH = RB // 16
    return r * RB + (((c ^ sw(r, D)) & (NCH - 1)) << 4)


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


@pytest.mark.parametrize("D", [32, 64, 128])
def test_row_reads_are_conflict_free(D):
    for row0 in (0, 32):
        for ks in range(D // 16):
            for h in (0, 1):                      # a ds_read_b128 lane group lies inside one 32-lane half
                for grp in B128_GROUPS:
                    slots = {(off(row0 + r, 2 * ks + h, D) % 256) // 16 for r in grp}
                    assert len(slots) == 16       # 16 lanes x 16 B = the 64 banks exactly once


@pytest.mark.parametrize("D", [32, 64, 128])
def test_transposed_reads_are_conflict_free(D):
    for k0 in (0, 16, 32, 48):
        for w in (0, 1):
            for dt in range(D // 32):
                for h in (0, 1):                  # banking of ds_read_b64_tr_b16 is per 32-lane half
                    banks = set()
                    for G in (2 * h, 2 * h + 1):
                        for q in range(4):
                            for p in range(4):
                                a = off(k0 + 8 * w + 4 * h + q, 4 * dt + 2 * (G & 1) + (p >> 1), D) + 8 * (p & 1)
                                banks.add((a % 256) // 8)
                    assert len(banks) == 32       # 32 lanes x 8 B = the 64 banks exactly once


@pytest.mark.parametrize("D", [32, 64, 128])
def test_swizzle_is_a_row_bijection_and_tile_offsets_are_immediates(D):
    nch = 2 * D // 16
    for r in range(64):
        offs = [off(r, c, D) for c in range(nch)]
        assert len(set(offs)) == nch and all(o // (2 * D) == r and o % 16 == 0 for o in offs)
    for k0 in (0, 16, 32, 48):                    # A2Offsets::trread + k0 * RB
        for w in (0, 1):
            for h in (0, 1):
                for q in range(4):
                    for c in range(nch):
                        assert off(k0 + 8 * w + 4 * h + q, c, D) == off(8 * w + 4 * h + q, c, D) + k0 * 2 * D
    for r in range(32):                           # A2Offsets::rowread + 32 * RB
        for c in range(nch):
            assert off(32 + r, c, D) == off(r, c, D) + 32 * 2 * D
