"""CPU: the restatement of the ε generator (tests/philox_ref.py) against Philox4x32-10's published known-answer vectors
(Random123's kat_vectors: counter, key → output)."""
import numpy as np

from tests import philox_ref as P

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_known_answer_vectors():
    for ctr, key, want in KAT:
        got = P.philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in got) == want


def test_words_are_the_blocks_in_order():
    w = P.words(10, seed=(0x299f31d0 << 32) | 0xa4093822, offset=(0x03707344 << 32) | 0x13198a2e, call=0x85a308d3)
    blk2 = P.philox4x32_10(2, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(v) for v in w[8:10]] == [int(blk2[0]), int(blk2[1])]
    assert np.array_equal(P.words(7, 5, 9, 1, epoch=3), P.words(7, 5, 12, 1))


def test_box_muller_moments():
    z = P.normals(P.words(1 << 18, seed=7, offset=0, call=0))
    assert abs(z.mean()) < 6e-3 and abs(z.var() - 1) < 1e-2 and abs((z ** 4).mean() - 3) < 0.08
