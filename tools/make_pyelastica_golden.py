#!/usr/bin/env python3
"""ONE COMMAND that pins the oracle ("parity unpinned" until this has run):

    python tools/make_pyelastica_golden.py            # in an environment where `import elastica` works
                                                      # (uv sync --frozen in the reference; Python >= 3.12)

imports the REFERENCE (`/root/reference/gym_softrobot`, pyelastica 1.0.0, uv.lock:845-860), runs
SoftPendulum-v0, SoftPendulum3D-v0, OctoArmSingle-v0 and OctoFlat-v0 (with --muscle-envs also the COOMM muscle arm,
OctoArmPush-v0 / -v1: SURVEY 8(f) N3) for seeds {0, 1, 42, 123} under a
fixed action script (the shape of the reference's tests/envs/test_determinism.py:12-54: reset(seed),
then env.step over a sampled action list) and writes tests/golden/pyelastica_<env>_seed<k>.npz:
rod state after 1 / 10 / 100 raw substeps, obs / reward / flags / time after every env.step, full rod
state after steps 1, 3, 10, 126 (tools/pyelastica_pin.py holds the layout).  The files are a few
hundred KB in total and are DATA: commit them; the reference's Python never travels to the GPU box.

Then:   python -m pytest tests/test_pyelastica_fixtures.py        (oracle vs fixtures, 1e-5)
        python tools/sweep_switches.py                           (which recalled details match)

    --source oracle [--flip name=value ...]    the same files produced by THIS repo's C oracle,
        optionally with recalled details flipped: how tests/test_switch_sweep.py proves that the
        sweep recovers what a fixture was generated with (the oracle in disguise), and how the
        fixture tests are exercised while no PyElastica fixture exists.  Never write these under
        the `pyelastica` prefix into tests/golden/ (refused).
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

import pyelastica_pin as pin  # noqa: E402


def parse_flip(items):
    sw = {}
    for it in items or []:
        k, v = it.split("=", 1)
        if k not in pin.SWITCHES:
            raise SystemExit(f"unknown switch {k!r}; have {sorted(pin.SWITCHES)}")
        proto = pin.SWITCHES[k][0]
        sw[k] = v if isinstance(proto, str) else type(proto)(eval(v, {"__builtins__": {}}, {}))  # "4/3" -> 1.333
    return sw


def versions():
    out = {"numpy": np.__version__}
    for mod in ("elastica", "numba", "gymnasium"):
        try:
            out[mod] = getattr(__import__(mod), "__version__", "?")
        except Exception as exc:  # noqa: BLE001
            out[mod] = f"unavailable ({type(exc).__name__})"
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--source", choices=["pyelastica", "oracle"], default="pyelastica")
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden"))
    ap.add_argument("--prefix", default=None, help="file prefix (default: pyelastica for the reference, oracle otherwise)")
    ap.add_argument("--envs", nargs="*", default=list(pin.ENVS))
    ap.add_argument("--muscle-envs", action="store_true",
                    help="also the six COOMM muscle envs, OctoArmPush-v0 / -v1, OctoArmPullWeight-v0, OctoCrawl-v0, OctoArmTwo-v0, OctoReach-v0 (SURVEY 8(f) N3; needs `import coomm` as well: "
                         "uv.lock:173-175).  Their fixtures are what decides pyelastica_pin.MUSCLE_SWITCHES")
    ap.add_argument("--seeds", nargs="*", type=int, default=list(pin.SEEDS))
    ap.add_argument("--steps", type=int, default=None, help="env.steps per case (default: the schedule's)")
    ap.add_argument("--flip", action="append", help="--source oracle only: name=value of a recalled detail")
    args = ap.parse_args(argv)
    if args.muscle_envs:
        args.envs = list(args.envs) + [e for e in pin.MUSCLE_ENVS if e not in args.envs]
        if args.source == "pyelastica":
            try:
                import coomm  # noqa: F401
            except ImportError as exc:
                raise SystemExit(f"`import coomm` failed ({exc}): the muscle envs need COOMM at the reference's git pin "
                                 "(uv.lock:173-175).  Nothing was written.")
    prefix = args.prefix or ("pyelastica" if args.source == "pyelastica" else "oracle")
    out = Path(args.out)
    if args.source != "pyelastica" and prefix == "pyelastica" and out.resolve() == (ROOT / "tests" / "golden").resolve():
        raise SystemExit("refusing to write oracle-made files under the pyelastica prefix into tests/golden/")
    sw = parse_flip(args.flip)
    if sw and args.source == "pyelastica":
        raise SystemExit("--flip applies to --source oracle")
    if args.source == "pyelastica":
        try:
            import elastica  # noqa: F401
        except ImportError as exc:
            raise SystemExit(f"`import elastica` failed ({exc}): run this where the reference's environment is "
                             "installed (uv sync --frozen in the reference checkout).  Nothing was written.")
    out.mkdir(parents=True, exist_ok=True)
    written = []
    for env_id in args.envs:
        for seed in args.seeds:
            drv = (pin.PyElasticaDriver(env_id, args.reference) if args.source == "pyelastica"
                   else pin.OracleDriver(env_id, sw))
            rec = pin.record_case(drv, seed, args.steps)
            drv.close()
            rec["versions"] = np.array(json.dumps(versions()))
            if args.source == "oracle":
                rec["switches"] = np.array(json.dumps(dict(pin.default_switches(), **sw)))
            f = out / pin.fixture_name(env_id, seed, prefix)
            np.savez_compressed(f, **rec)
            written.append(f.name)
            print(f"{f.name}: {len(rec['obs'])} env.steps, final time {float(rec['time'][-1]):.6f}, "
                  f"terminated {bool(rec['terminated'].any())}, truncated {bool(rec['truncated'].any())}", flush=True)
    (out / f"{prefix}_README.json").write_text(json.dumps({
        "made_by": "tools/make_pyelastica_golden.py --source " + args.source,
        "versions": versions(), "files": written, "seeds": args.seeds, "envs": args.envs,
        "flipped": sw or None,
        "note": "inputs (stored actions, seeds) and expected outputs only; no reference source text",
    }, indent=1) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
