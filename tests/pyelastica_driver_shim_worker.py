"""Worker of tests/test_pyelastica_fixtures.py::test_pyelastica_driver_runs_the_reference_env_code:
runs tools/pyelastica_pin.PyElasticaDriver — the driver that will record the pin where pyelastica is
installed — over tools/refshim.py's stand-ins: the reference's REAL env classes (reset, set_action,
step, get_state, the attributes the driver reads) with a scripted stepper that integrates nothing.
Proves the generator's reference-facing code path end to end; the numbers it records are meaningless
(no physics) and are never written under tests/golden/.  Needs /root/reference (build container only)."""
import sys
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

import refshim  # noqa: E402

refshim.install()
import pyelastica_pin as pin  # noqa: E402


class ShimDriver(pin.PyElasticaDriver):
    def _post_reset(self):
        e = self.env
        # a straight unit-tangent rod so that get_state's arctan / means are finite
        rods = e.shearable_rods if self.octo else [e.shearable_rod]
        for i, r in enumerate(rods):
            r.tangents[0, :] = 0.6
            r.tangents[1, :] = 0.8
            r.mass[:] = 1.0
            ang = 2 * np.pi * i / max(len(rods), 1)          # arms radiating from the head: no two of them overlap
            s_ = np.linspace(0.04, 0.39, r.position_collection.shape[1])
            r.position_collection[0], r.position_collection[1] = np.cos(ang) * s_, np.sin(ang) * s_
        e.simulator._script = lambda calls, t, dt: t + dt          # the stepper advances the clock only


def main(out):
    warnings.simplefilter("ignore")
    shapes = {}
    for env_id in pin.ENVS:
        drv = ShimDriver(env_id)
        rec = pin.record_case(drv, 1, n_steps=2)
        drv.close() if hasattr(drv.env, "close") and not drv.octo else None
        shapes[env_id] = {k: list(np.shape(v)) for k, v in rec.items()}
        shapes[env_id]["_time"] = [float(t) for t in rec["time"]]
        shapes[env_id]["_source"] = str(rec["source"])
    import json

    Path(out).write_text(json.dumps(shapes))


if __name__ == "__main__":
    main(sys.argv[1])
