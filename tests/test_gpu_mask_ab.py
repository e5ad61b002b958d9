"""A/B control of the EXEC masks: the substep loops of the shipped library run with the
idle lanes / ghost slots switched off (SOFTROD_PLANAR_EXEC_MASK, SOFTROD_IDLE_LANES_EXEC_MASK,
SOFTROD_OCTO_GHOST_MASK), which is lane-divergent control flow around barriers and cross-lane reads
and rests on the invariants written at the masks (softrod_fast.hpp general_substeps).  The `nomask`
build (csrc/Makefile) is the same source with every mask off: both must agree BIT FOR BIT on every
workload, so a compiler or ISA change that breaks an invariant fails here."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_masked_and_unmasked_builds_are_bit_identical(hip_lib, tmp_path):
    nomask = ROOT / "variants" / "libsoftrod_nomask.so"
    if not nomask.exists():
        subprocess.run(["make", "-C", str(ROOT / "gym_softrobot_amd" / "csrc"), "nomask"], check=True)
    outs = {}
    for tag, lib in (("masked", None), ("nomask", nomask)):
        env = dict(os.environ)
        env.pop("SOFTROD_HIP_LIB", None)
        if lib is not None:
            env["SOFTROD_HIP_LIB"] = str(lib)
        out = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, str(ROOT / "tests" / "mask_ab_worker.py"), str(out)], check=True, env=env,
                       timeout=600)
        outs[tag] = np.load(out)
    assert str(outs["nomask"]["library"]).endswith("libsoftrod_nomask.so")
    assert str(outs["masked"]["library"]).endswith("libsoftrod_hip.so")
    keys = [k for k in outs["masked"].files if k != "library"]
    assert len(keys) >= 5 * 6
    for k in keys:
        a, b = outs["masked"][k], outs["nomask"][k]
        assert np.array_equal(a, b, equal_nan=True), f"{k}: masked and unmasked builds differ"
