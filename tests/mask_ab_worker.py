"""Worker of tests/test_gpu_mask_ab.py: steps four workloads on whichever libsoftrod_hip build
SOFTROD_HIP_LIB names and dumps outputs and rod states into argv[1] (.npz)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main(out):
    import torch

    import gym_softrobot_amd as gsa

    dump = {"library": np.array(str(gsa._capi.library_path()))}
    cases = [("SoftPendulum-v0", 8, {}, 22.0, 3), ("SoftPendulum3D-v0", 4, {}, 1.0, 2),
             ("OctoArmSingle-v0", 4, {}, 6.0, 1), ("OctoArmSingle-v0", 4, {"n_elems": 100}, 6.0, 1),
             ("OctoFlat-v0", 4, {}, 22.0, 1)]
    for ci, (name, n, kw, amax, steps) in enumerate(cases):
        env = gsa.make_vec(name, n, **kw)
        env.reset(seed=3)
        rng = np.random.default_rng(11 + ci)
        for t in range(steps):
            a = rng.uniform(-amax, amax, (n, env.action_dim)).astype(np.float32)
            obs, rew, te, tr, _ = env.step(a)
        torch.cuda.synchronize()
        tag = f"{ci}_{name}"
        dump[tag + "_obs"] = obs.cpu().numpy()
        dump[tag + "_rew"] = rew.cpu().numpy()
        # the slots that carry a node / element (idle lanes and ghost slots hold no state)
        st = env.backend.octo_state_numpy() if env.backend.is_octo else env.backend.state_numpy()
        for k in ("x", "v", "w", "Q") + (("head_x", "head_v", "head_Q", "head_w") if env.backend.is_octo else ()):
            dump[tag + "_" + k] = st[k]
        env.close()
    np.savez(out, **dump)


if __name__ == "__main__":
    main(sys.argv[1])
