"""The bench's numpy input generator against the oracle's and the golden raw checksums."""
import numpy as np

from opengpc_amd.synth import synth_batch, synth_pair


def test_synth_matches_golden_and_oracle(oracle, golden):
    for c in golden["cases"]:
        L, R = synth_pair(c["W"], c["H"], c["s"], c["D"])
        assert ["%016x" % oracle.fnv(L), "%016x" % oracle.fnv(R)] == c["raw"]
    for s, D in [(1, 40), (7, 15), (300, 8 + 300 % 64)]:
        L, R = synth_pair(160, 48, s, D)
        Lo, Ro = oracle.synth_pair(160, 48, s, D)
        assert np.array_equal(L, Lo) and np.array_equal(R, Ro)
    Lb, Rb = synth_batch(96, 32, [5, 70])
    assert np.array_equal(Lb[1], oracle.synth_pair(96, 32, 70, 8 + 6)[0])
