"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and
exports every entry point include/softrod.h declares; the ctypes mirror of
softrod_config is layout-identical to the C struct; calls fail loudly (no fallback)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from gym_softrobot_amd import _capi

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "softrod.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(softrod_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(hip_lib):
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/softrod.h but not exported"
    assert set(names) == set(_capi.EXPORTED_SYMBOLS)


def test_config_layout_and_defaults_match_c(hip_lib):
    c = _capi.SoftrodConfig()
    assert hip_lib.softrod_config_softpendulum(C.byref(c), 7) == 0
    py = _capi.softpendulum_config(7)
    assert c.struct_size == C.sizeof(_capi.SoftrodConfig)
    assert bytes(c) == bytes(py)
    assert py.n_substeps == 400 and py.n_elem == 50      # soft_pendulum.py:64,78
    assert py.shear_modulus == pytest.approx(1e6 / 3.0)


def test_every_python_config_builder_matches_its_c_twin(hip_lib):
    pairs = [("softrod_config_softpendulum", _capi.softpendulum_config),
             ("softrod_config_softpendulum3d", _capi.softpendulum3d_config),
             ("softrod_config_arm_single", _capi.arm_single_config),
             ("softrod_config_octo_flat", _capi.octo_flat_config)]
    for cname, pyfn in pairs:
        c = _capi.SoftrodConfig()
        assert getattr(hip_lib, cname)(C.byref(c), 5) == 0
        assert bytes(c) == bytes(pyfn(5)), cname
        assert hip_lib.softrod_config_action_dim(C.byref(c)) == _capi.config_action_dim(c)
        assert hip_lib.softrod_config_obs_dim(C.byref(c)) == _capi.config_obs_dim(c)


def test_create_rejects_bad_config_and_never_falls_back(hip_lib):
    h = C.c_void_p()
    cfg = _capi.softpendulum_config(4)
    cfg.struct_size = 8
    assert hip_lib.softrod_create(C.byref(cfg), 0, C.byref(h)) == -1
    cfg = _capi.softpendulum_config(4)
    cfg.n_elem = 127  # one rod per 64-lane wavefront, two nodes per lane: at most 126 elements
    assert hip_lib.softrod_create(C.byref(cfg), 0, C.byref(h)) == -1
    cfg.n_elem = 64   # two nodes per lane exist for the fast kernel only
    cfg.math_mode = _capi.MATH_LIBM
    assert hip_lib.softrod_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert b"SOFTROD_MATH_FAST" in hip_lib.softrod_last_error(None)
    assert not h.value
    # the fast kernels linearise theta / sin(theta + eps_sin) in eps_sin; a config where that
    # does not hold (acos_shift = 0: a straight joint would flip the sign of the bending stiffness) is
    # refused in fast mode — before any device is looked for — and left to the libm kernel
    for field, value in (("acos_shift", 0.0), ("eps_sin", 1e-6), ("eps_sin", -1e-14)):
        cfg = _capi.softpendulum_config(4)
        setattr(cfg, field, value)
        assert hip_lib.softrod_create(C.byref(cfg), 0, C.byref(h)) == -1, (field, value)
        assert b"eps_sin" in hip_lib.softrod_last_error(None) and not h.value


def test_backend_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gym_softrobot_amd as gsa

    with pytest.raises(_capi.SoftrodError):
        gsa.make_vec("SoftPendulum-v0", 2)
    with pytest.raises(_capi.SoftrodError):
        gsa.make("SoftPendulum-v0")


def test_library_is_the_build_of_the_sources_on_disk(hip_lib):
    """softrod_source_hash(): the .so that will travel to the GPU box must be the build of the
    sources in the tree (a stale binary would be benchmarked against the wrong profile tables and
    tested against the wrong kernels)."""
    assert _capi.library_source_hash() == _capi.source_hash(), \
        "libsoftrod_hip.so is older than csrc/ or include/: run `python __graft_entry__.py build`"
    assert re.fullmatch(r"[0-9a-f]{16}", _capi.library_source_hash())


def test_profile_tables_name_the_build_they_were_measured_on():
    """Every entry of the tables bench.py prices kernels against carries the source hash of the
    library it was measured on (tools/update_profile_tables.py); bench.py withholds `frac` when it
    is not the loaded library's."""
    import json
    import sys

    sys.path.insert(0, str(ROOT))
    import bench

    for name in ("valu_counts.json", "hbm_traffic.json"):
        doc = json.loads((ROOT / "profiles" / name).read_text())
        entries = {k: v for k, v in doc.items() if not k.startswith("_")}
        assert len(entries) == 11, name       # six workloads + the tapered arm; round 6: the libm kernel, OctoArmPush-v1, OctoArmPullWeight-v0, OctoCrawl-v0
        for k, v in entries.items():
            assert re.fullmatch(r"[0-9a-f]{16}", v["source_hash"]), (name, k)
            assert (ROOT / v["source"].split(" ")[0]).exists(), (name, k, v["source"])
    rec = {"source_hash": "0123456789abcdef", "valu_instr_per_rod_substep": 1.0}
    assert bench.fresh_or_none(rec, "0123456789abcdef") == (rec, None)
    stale, why = bench.fresh_or_none(rec, "fedcba9876543210")
    assert stale is None and "re-run" in why
    assert bench.fresh_or_none(None, "0123456789abcdef")[0] is None


def test_docs_quote_the_header_abi_version():
    """INTEGRATION.md's binding snippet asserts the ABI version of the header it was written
    against (it said 11 under a header at 13 in round 3)."""
    header = (ROOT / "include" / "softrod.h").read_text()
    ver = int(re.search(r"#define SOFTROD_ABI_VERSION (\d+)", header).group(1))
    doc = (ROOT / "INTEGRATION.md").read_text()
    assert int(re.search(r"softrod_abi_version\(\) == (\d+)", doc).group(1)) == ver
    assert f"ABI v{ver}" in doc
    from gym_softrobot_amd import _capi

    assert _capi.ABI_VERSION == ver
    import ctypes

    assert f"{ctypes.sizeof(_capi.SoftrodConfig)} bytes" in doc
