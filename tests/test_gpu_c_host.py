"""The C-ABI is self-contained: a plain C host (examples/c_host.c: HIP runtime + libsoftrod_hip.so,
no Python, no torch) must produce what the Python host path produces, bit for bit."""
import math
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_c_host_matches_python_host(hip_lib, tmp_path):
    import torch

    assert torch.cuda.is_available()
    exe = tmp_path / "c_host"
    csrc = ROOT / "gym_softrobot_amd" / "csrc"
    subprocess.run(["gcc", "-std=gnu11", "-o", str(exe), str(ROOT / "examples" / "c_host.c"),
                    f"-I{ROOT / 'include'}", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    f"-L{csrc}", "-lsoftrod_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    n, steps = 4, 3
    out = subprocess.run([str(exe), str(n), str(steps)], check=True, capture_output=True, text=True, timeout=300).stdout
    rows = [l.split() for l in out.strip().splitlines()]
    assert len(rows) == n + n * steps

    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    be = HipRodBackend(_capi.softpendulum_config(n), device=0)
    theta0 = np.array([(90.0 + (0.1 * i - 0.2) * 10.0) * math.pi / 180.0 for i in range(n)])
    be.reset(theta0)
    obs = be.observe(None).cpu().numpy()
    for i in range(n):
        np.testing.assert_array_equal(np.array(rows[i][2:6], np.float32), obs[i])
    k = n
    for t in range(steps):
        a = np.array([np.float32(22.0 * math.sin(1.0 + 0.7 * i + 1.3 * t)) for i in range(n)], np.float32)
        o, r, te, tr = (x.cpu().numpy() for x in be.step(a))
        for i in range(n):
            row = rows[k]
            k += 1
            assert row[0] == "step" and int(row[1]) == t and int(row[2]) == i
            np.testing.assert_array_equal(np.array(row[3:7], np.float32), o[i])
            assert float(row[7]) == r[i] and int(row[8]) == te[i] and int(row[9]) == tr[i]
    be.close()
