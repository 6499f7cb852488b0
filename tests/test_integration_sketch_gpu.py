"""INTEGRATION.md section B, executed: the ~20-line ctypes stub a maintainer would add to the reference
(`vmas/simulator/native.py`, class NativeStep) is taken VERBATIM from the document, pointed at the in-tree library,
and must step a reference world exactly like the package's own host side does."""
import os
import re

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.reference]

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _sketch():
    from vectorizedmultiagentsimulator_amd import _abi

    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## B."):md.index("## C.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    src = next(b for b in blocks if "class NativeStep" in b)
    assert 'C.CDLL("libvmas_hip.so")' in src
    ns = {}
    exec(src.replace('C.CDLL("libvmas_hip.so")', f"C.CDLL({_abi.LIB_PATH!r})"), ns)
    return ns["NativeStep"]


@pytest.mark.parametrize("scenario,kw", [("balance", dict(n_agents=3)), ("transport", {})])
def test_the_documented_binding_steps_a_reference_world(scenario, kw):
    from oracle import ref
    from ref_backend import pack_ft, pack_state
    from vectorizedmultiagentsimulator_amd.backend import HipWorld
    from vectorizedmultiagentsimulator_amd.spec import spec_from_world

    NativeStep = _sketch()
    B = 200
    env = ref.make_env(scenario, num_envs=B, device="cuda:0", seed=0, **kw)
    env.step([(torch.rand(B, a.action_size, device="cuda:0") * 2 - 1) * 0.5 for a in env.agents])  # forces are set
    world = env.world
    spec = spec_from_world(world)
    cd = spec.to_ctypes()  # VmasWorldDesc + the arrays it points to (kept alive by `cd`)
    nat = NativeStep(world, cd.world)
    st, ft = torch.from_numpy(pack_state(world)).cuda(), torch.from_numpy(pack_ft(world)).cuda()
    nat.state[:, :, :B].copy_(st)
    nat.agent_ft[: ft.shape[0], :, :B].copy_(ft)
    hw = HipWorld(spec, B, "cuda:0")
    hw.state.copy_(nat.state)
    hw.agent_ft.copy_(nat.agent_ft)
    for _ in range(3):
        nat.step()
        hw.step()
    torch.cuda.synchronize()
    assert torch.equal(nat.state.view(torch.int32), hw.state.view(torch.int32))
    assert not torch.equal(nat.state[:, :, :B], st)  # it did step
