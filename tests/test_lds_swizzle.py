"""Bank-conflict check of the LDS address swizzle of the split transform (csrc/fbs_ntt_split.hpp `phys`), by
enumeration with the gfx950 banking rules of MI355X_MICROARCH.md (8-byte reads: two 32-lane groups over 64 four-byte
banks; 8-byte writes: four 16-lane groups over 32 banks).  Pure host arithmetic: the formula is restated here and
compared with the header's text so the two cannot drift apart."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def phys(i):
    return i ^ ((i >> 4) & 7) ^ (((i >> 6) & 3) << 3)


def local_index(lo, t, r):
    return ((t >> lo) << (lo + 3)) | (r << lo) | (t & ((1 << lo) - 1))


def test_header_uses_this_formula():
    text = open(os.path.join(ROOT, "tfhe_fbs_map_amd", "csrc", "fbs_ntt_split.hpp")).read()
    m = re.search(r"uint32_t phys\(uint32_t i\) \{ return (.*?); \}", text)
    assert m and m.group(1).replace("u", "") == "i ^ ((i >> 4) & 7) ^ (((i >> 6) & 3) << 3)"


def test_swizzle_is_a_permutation_and_xor_linear():
    assert sorted(phys(i) for i in range(512)) == list(range(512))
    for a in (1, 8, 64, 77, 300, 511):
        for b in (2, 16, 128, 255, 448):
            assert phys(a ^ b) == phys(a) ^ phys(b)


def test_every_exchange_pattern_is_conflict_free():
    for lo in (6, 3, 0):                      # the three register/lane layouts of a half transform
        for r in range(8):
            words = [phys(local_index(lo, t, r)) for t in range(64)]
            for g in range(2):                # ds_read_b64: 32 lanes share 64 banks = 32 eight-byte positions
                assert len({w % 32 for w in words[32 * g:32 * g + 32]}) == 32, (lo, r, "read")
            for g in range(4):                # ds_write_b64: 16 lanes share 32 banks = 16 eight-byte positions
                assert len({w % 16 for w in words[16 * g:16 * g + 16]}) == 16, (lo, r, "write")


# ---- the one exchange of the 256-point lane-transposition transform (csrc/fbs_ntt_lane.hpp) ----------------------------
def lane_phys(j):
    return j ^ ((j >> 4) & 15) ^ (((j >> 6) & 1) << 4)


def lane_index_b(ln, m):
    return ((ln >> 4) << 6) | (m << 4) | (ln & 15)


def lane_index_c(ln, m):
    return ((ln & 15) << 4) | (m << 2) | (ln >> 4)


def test_lane_header_uses_these_formulas():
    text = open(os.path.join(ROOT, "tfhe_fbs_map_amd", "csrc", "fbs_ntt_lane.hpp")).read()
    m = re.search(r"uint32_t phys\(uint32_t j\) \{ return (.*?); \}", text)
    assert m and m.group(1).replace("u", "") == "j ^ ((j >> 4) & 15) ^ (((j >> 6) & 1) << 4)"
    assert "return ((ln >> 4) << 6) | ((uint32_t)m << 4) | (ln & 15u);" in text
    assert "return ((ln & 15u) << 4) | ((uint32_t)m << 2) | (ln >> 4);" in text


def test_lane_exchange_is_a_conflict_free_permutation():
    assert sorted(lane_phys(j) for j in range(256)) == list(range(256))
    for a in (1, 8, 64, 77, 200, 255):
        for b in (2, 16, 128, 99):
            assert lane_phys(a ^ b) == lane_phys(a) ^ lane_phys(b)
    for layout in (lane_index_b, lane_index_c):
        assert sorted(layout(ln, m) for ln in range(64) for m in range(4)) == list(range(256))
        for m in range(4):
            words = [lane_phys(layout(ln, m)) for ln in range(64)]
            for g in range(2):                # read 32 lanes at a time over 32 eight-byte positions
                assert len({w % 32 for w in words[32 * g:32 * g + 32]}) == 32, (layout.__name__, m, "read")
            for g in range(4):                # written 16 lanes at a time over 16
                assert len({w % 16 for w in words[16 * g:16 * g + 16]}) == 16, (layout.__name__, m, "write")


# ---- ... and of the 512-point one (LaneNtt512) -------------------------------------------------------------------------
def lane512_phys(j):
    return j ^ ((j >> 5) & 15) ^ (((j >> 4) & 1) << 3) ^ (((j >> 7) & 1) << 4)


def lane512_index_b(ln, m):
    return ((ln >> 4) << 7) | ((m & 1) << 6) | ((m >> 2) << 5) | (((m >> 1) & 1) << 4) | (ln & 15)


def lane512_index_c(ln, m):
    return ((ln & 31) << 4) | (m << 1) | (ln >> 5)


def test_lane512_header_uses_these_formulas():
    text = open(os.path.join(ROOT, "tfhe_fbs_map_amd", "csrc", "fbs_ntt_lane.hpp")).read()
    assert "return j ^ ((j >> 5) & 15u) ^ (((j >> 4) & 1u) << 3) ^ (((j >> 7) & 1u) << 4);" in text
    assert "return ((ln >> 4) << 7) | ((uint32_t)(m & 1) << 6) | ((uint32_t)(m >> 2) << 5) | ((uint32_t)((m >> 1) & 1) << 4) | (ln & 15u);" in text
    assert "return ((ln & 31u) << 4) | ((uint32_t)m << 1) | (ln >> 5);" in text


def test_lane512_exchange_is_a_conflict_free_permutation():
    assert sorted(lane512_phys(j) for j in range(512)) == list(range(512))
    for a in (1, 8, 64, 77, 300, 511):
        for b in (2, 16, 128, 255, 448):
            assert lane512_phys(a ^ b) == lane512_phys(a) ^ lane512_phys(b)
    for layout in (lane512_index_b, lane512_index_c):
        assert sorted(layout(ln, m) for ln in range(64) for m in range(8)) == list(range(512))
        for ln in (0, 5, 37, 63):                                        # the kernel adds lane and register parts with XOR
            for m in range(8):
                assert layout(ln, m) == layout(ln, 0) ^ layout(0, m)
        for m in range(8):
            words = [lane512_phys(layout(ln, m)) for ln in range(64)]
            for g in range(2):
                assert len({w % 32 for w in words[32 * g:32 * g + 32]}) == 32, (layout.__name__, m, "read")
            for g in range(4):
                assert len({w % 16 for w in words[16 * g:16 * g + 16]}) == 16, (layout.__name__, m, "write")
