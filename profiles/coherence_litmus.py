"""Litmus for the accumulator access pattern of the multi-kernel Krylov solvers (fg_coherence_litmus).

    python profiles/coherence_litmus.py [iterations] [number of shapes]

For a few (systems, cells) shapes -- the Airfoil2D batch the defect was captured on among them -- runs the five-launch recurrence
skeleton with plain loads / stores of the sum slots (the round-1 pattern) and with agent-scope atomic loads / stores (what the
solvers use now) and prints, per slot, how many reads did not return the full sum.  One JSON line per run."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidgym_amd import _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
lib = L.load()
torch.cuda.init()
torch.zeros(1, device="cuda")
SLOTS = {0: "rho0", 1: "rho1", 2: "rw.v", 3: "ss", 4: "ts", 5: "tt", 6: "rr", 11: "flag"}
NAMES = {0: "plain load, plain store (round 1)", 1: "atomic load, plain store", 10: "plain load, atomic store", 20: "plain load, atomic exchange",
         11: "atomic load, atomic store", 21: "atomic load, atomic exchange"}
shapes = ((32, 46664), (16, 46664), (48, 11776), (3, 262144), (256, 2048))
if len(sys.argv) > 2:
    shapes = shapes[:int(sys.argv[2])]
for nsys, cells in shapes:
    for atomic in (0, 1, 10, 20, 11, 21):
        bad = (ctypes.c_int64 * 12)()
        val = (ctypes.c_double * 12)()
        t0 = time.time()
        L.check(lib.fg_coherence_litmus(atomic, nsys, cells, iters, bad, val, None))
        dt = time.time() - t0
        print(json.dumps({"access": NAMES[atomic], "systems": nsys, "cells": cells, "iterations": iters,
                          "launches": 5 * iters, "us_per_launch": round(dt / (5 * iters) * 1e6, 2),
                          "bad_reads": {SLOTS[k]: int(bad[k]) for k in SLOTS if bad[k]},
                          "first_bad_value": {SLOTS[k]: float(val[k]) for k in SLOTS if bad[k]}}), flush=True)
