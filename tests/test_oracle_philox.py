"""Known-answer vectors of Philox4x32-10 (Random123 distribution, examples/kat_vectors) pin oracle/philox.py; the device
kernels are compared with the oracle in the -m gpu / emulator tests."""
import numpy as np

from oracle import philox


def test_philox4x32_10_known_answers():
    kat = [((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = philox.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert tuple(int(x) for x in got) == out, [hex(int(x)) for x in got]


def test_action_noise_is_standard_normal_and_counter_based():
    n = philox.action_noise(12345, np.arange(200000), 6, stream=1)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1.0) < 0.01
    assert abs(np.corrcoef(n[:, 0], n[:, 1])[0, 1]) < 0.01 and abs(np.corrcoef(n[:-1, 0], n[1:, 0])[0, 1]) < 0.01
    np.testing.assert_array_equal(philox.action_noise(12345, [77, 5], 6, 1), n[[77, 5]])       # any subset, any order
    assert not np.array_equal(philox.action_noise(12345, [77], 6, 0), n[[77]])                 # streams differ
