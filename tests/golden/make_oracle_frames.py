#!/usr/bin/env python3
"""Freezes the CPU oracle's output for a handful of named acquisitions into
tests/golden/oracle_frames.npz (a regression anchor for the checker itself: the oracle is the
yardstick every GPU parity test uses, so a change to it must be deliberate and visible).

These are outputs of THIS repository's oracle (oracle/*.c), not of the reference: the fixture
pins the oracle against drift, the reference pins are tests/golden/host_math.npz and
shm_layout.txt.

    python tests/golden/make_oracle_frames.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

NAMES = ["config1_small", "config2_small", "config4_small", "config5_small", "config5_literal_order", "forces",
         "uforces_sparse", "readi", "hercules_order12", "rca_vls_cw", "rca_a1s2", "hercules_chirp"]


def main():
    from oracle import binding as oracle
    from tests import cases
    out = {}
    for name in NAMES:
        acq = cases.make(name)
        frame, pairs = oracle.beamform(acq.bp, acq.rf, acq.filters)
        out[name] = frame[::8] if frame.shape[0] > 16 else frame       # every 8th z-plane of the larger volumes
        out[name + ".pairs"] = np.array(pairs, np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **out)
    print("wrote", len(NAMES), "frames,", os.path.getsize(os.path.join(HERE, "oracle_frames.npz")), "bytes")


if __name__ == "__main__":
    main()
