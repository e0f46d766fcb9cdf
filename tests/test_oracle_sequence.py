"""The tracking chain of tests/tracking_chain.py through the CPU oracle alone: it tracks the ground truth, and it is
sensitive to the pose it is handed -- the fact that bounds what sequence-level parity can mean (tests/test_gpu_sequence.py)."""
import numpy as np

import tracking_chain as tc
from android_svo_amd import synth


def test_oracle_chain_tracks_and_amplifies_disturbances():
    seq = tc.make_sequence(n_frames=20)
    st = tc.OracleStages(seq)
    base, n_base, _ = tc.run_chain(seq, st, 2)
    truth = seq["truth"][1:]
    err = np.array([synth.pose_error(a, t) for a, t in zip(base, truth)])
    assert err[:, 0].max() < 2e-3 and err[:, 1].max() < 5e-3                  # sub-pixel tracking, no drift
    assert all(n > tc.MAX_FTS for n in n_base)                                # the cell loop stops just past Config::maxFts()
    # the same chain with the handed-over pose disturbed by 1e-15 m after every frame
    pert, _, _ = tc.run_chain(seq, st, 2, perturb_each_frame=1e-15)
    d = np.array([synth.pose_error(a, b)[1] for a, b in zip(base, pert)])
    assert d[:5].max() < 1e-13                                                # early frames: the chain contracts
    assert d[-1] > 100 * 1e-15                                                # later frames: gain > 1 per frame
    assert d.max() < 1e-6                                                     # still far inside the north_star tolerance here
