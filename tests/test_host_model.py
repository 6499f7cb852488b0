"""Host-side object model (core.py) and scenarios - runs on CPU tensors, no GPU."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import load
from vectorizedmultiagentsimulator_amd import core
from vectorizedmultiagentsimulator_amd.scenarios.balance import Scenario as Balance
from vectorizedmultiagentsimulator_amd.spec import WorldSpec, spec_from_world


def test_balance_world_spec_equals_reference_extracted_spec():
    """Our own scenario builds the same static world the reference does: the spec
    extracted from it is identical to the one the golden generator extracted from the
    reference's live World (tests/golden/balance_n*.npz)."""
    for n, fixture in ((3, "balance_n3"), (4, "balance_n4")):
        sc = Balance()
        w = sc.env_make_world(8, "cpu", n_agents=n)
        assert json.loads(w.spec.to_json()) == json.loads(load(fixture).spec.to_json())


def test_spec_json_roundtrip():
    g = load("waterfall")
    assert json.loads(WorldSpec.from_json(g.spec.to_json()).to_json()) == json.loads(g.spec.to_json())


def test_state_views_alias_the_packed_buffer():
    sc = Balance()
    w = sc.env_make_world(5, "cpu", n_agents=3)
    sc.env_reset_world_at(None)
    line = [e for e in w.entities if e.name == "line"][0]
    i = w.entities.index(line)
    assert line.state.pos.shape == (5, 2) and line.state.rot.shape == (5, 1)
    line.state.pos = torch.full((5, 2), 0.25)
    assert torch.all(w._state[i, 0:2, :5] == 0.25)
    line.state.pos[2] = torch.tensor([1.0, 2.0])  # in-place indexed write (core.py:759-760)
    assert w._state[i, 0, 2] == 1.0 and w._state[i, 1, 2] == 2.0
    line.set_pos(torch.tensor([0.5, -0.5]), batch_index=None)  # broadcast over the batch
    assert torch.all(w._state[i, 0, :5] == 0.5) and torch.all(w._state[i, 1, :5] == -0.5)
    line.set_rot(torch.tensor([0.3]), batch_index=1)
    assert w._state[i, 4, 1] == pytest.approx(0.3)
    with pytest.raises(AssertionError):
        line.state.pos = torch.zeros(4, 2)  # wrong batch dim (core.py:224-233)
    agent = w.agents[1]
    agent.state.force = torch.ones(5, 2)
    assert torch.all(w._agent_ft[1, 0:2, :5] == 1.0)
    w.reset(env_index=3)
    assert torch.all(w._state[:, :, 3] == 0) and torch.all(w._agent_ft[:, :, 3] == 0)
    assert w._state[i, 0, 0] == 0.5  # other envs untouched


def test_balance_reset_distribution_matches_reference_layout():
    torch.manual_seed(0)
    sc = Balance()
    w = sc.env_make_world(2048, "cpu", n_agents=4)
    sc.env_reset_world_at(None)
    ents = {e.name: e for e in w.entities}
    line, pkg, floor = ents["line"].state.pos, ents["package"].state.pos, ents["floor"].state.pos
    assert torch.allclose(line[:, 1], torch.full((2048,), -1 + 0.06))  # balance.py:119-124
    assert line[:, 0].min() >= -0.6 and line[:, 0].max() <= 0.6
    assert torch.allclose(pkg[:, 1] - line[:, 1], torch.full((2048,), 0.05))
    assert (pkg[:, 0] - line[:, 0]).abs().max() <= 0.4 - 0.05 + 1e-6
    assert torch.all(floor[:, 0] == 0) and torch.allclose(floor[:, 1], torch.full((2048,), -1 - 0.5 - 0.03))
    a0, a3 = ents["agent_0"].state.pos, ents["agent_3"].state.pos
    assert torch.allclose(a3[:, 0] - a0[:, 0], torch.full((2048,), 0.8 - 0.03), atol=1e-6)
    assert torch.allclose(a0[:, 1], line[:, 1] - 0.06)
    # partial reset touches one env only
    before = w._state.clone()
    sc.env_reset_world_at(7)
    diff = (w._state != before).any(dim=0).any(dim=0)
    assert diff[7] and diff.sum() == 1


def test_entities_cannot_be_added_after_state_exists():
    w = core.World(4, "cpu")
    w.add_agent(core.Agent("a"))
    w.reset(None)
    with pytest.raises(AssertionError):
        w.add_landmark(core.Landmark("l"))


def test_joint_builds_link_landmark_and_constraints():
    w = core.World(3, "cpu", substeps=2)
    a, b = core.Agent("a"), core.Agent("b")
    w.add_agent(a)
    w.add_agent(b)
    j = core.Joint(a, b, anchor_a=(0, 0), anchor_b=(0, 0), dist=0.2, rotate_a=False, rotate_b=True, collidable=True,
                   width=0.05)
    w.add_joint(j)
    spec = spec_from_world(w)
    assert [e.name for e in spec.entities] == ["joint a b", "a", "b"]
    assert len(spec.joints) == 2 and spec.joints[0].per_env_fixed_rotation and not spec.joints[1].per_env_fixed_rotation
    a.set_pos(torch.tensor([0.0, 0.0]), None)
    b.set_pos(torch.tensor([0.2, 0.0]), None)  # notify() re-places the link between them
    assert torch.allclose(j.landmark.state.pos, torch.tensor([[0.1, 0.0]] * 3))
    with pytest.raises(AssertionError):  # joints need substeps > 1 (core.py:1167)
        core.World(3, "cpu").add_joint(core.Joint(core.Agent("x"), core.Agent("y")))


def test_generated_specialisation_is_current():
    """csrc/vmas_spec_gen.h (the world-specialised kernel's tables) is what scripts/gen_spec.py generates from the current
    planner and scenario: a change to either must be followed by re-running the script (the library would otherwise -
    correctly, but silently - fall back to the interpreter)."""
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    rc = subprocess.run([sys.executable, os.path.join(root, "scripts", "gen_spec.py"), "--check"], capture_output=True, text=True)
    assert rc.returncode == 0, "csrc/vmas_spec_gen.h is stale: run python scripts/gen_spec.py and rebuild\n" + rc.stderr[-2000:]


def test_planning_world_matches_the_generated_tables():
    """A planning world (device -1: no GPU touched) of balance n_agents=4 at 32768 environments reports the specialisation
    as matching its schedule, and cannot be stepped."""
    import ctypes as C

    from vectorizedmultiagentsimulator_amd import _abi as A
    from vectorizedmultiagentsimulator_amd.scenarios.balance import Scenario

    w = Scenario().env_make_world(32768, "cpu", n_agents=4)
    cd = w.spec.to_ctypes()
    lib = A.load_library()
    lib.vmas_debug_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    h = C.c_void_p()
    assert lib.vmas_world_create(C.byref(cd.world), 32768, -1, C.byref(h)) == 0, A.last_error()
    try:
        assert lib.vmas_world_reserve_epilogue(h, *w.epilogue_hint) == 0
        meta = (C.c_int32 * 24)()
        assert lib.vmas_debug_schedule(h, None, 0, meta) == 0
        assert meta[0] == 8 and meta[23] == 0, list(meta)  # 8 waves per tile, specialisation 0 (SpecBalance4)
        assert lib.vmas_world_get_specialized(h) == 1
        assert lib.vmas_world_step(h, C.c_void_p(64), C.c_void_p(64), 32768, None, None) != 0
        assert b"planning world" in lib.vmas_last_error()
    finally:
        lib.vmas_world_destroy(h)


def test_lidar_kernel_choice_validates_its_mode_without_a_gpu():
    """vmas_world_set_lidar_compact on a planning world: the mode is checked (-1 library's choice, 0 plain kernel, 1 the
    lane-compacted cast), and a world without a registered sphere-only sensor set never reports the compacted form."""
    import ctypes as C

    from vectorizedmultiagentsimulator_amd import _abi as A
    from vectorizedmultiagentsimulator_amd.scenarios.balance import Scenario

    w = Scenario().env_make_world(4096, "cpu", n_agents=3)
    cd = w.spec.to_ctypes()
    lib = A.load_library()
    h = C.c_void_p()
    assert lib.vmas_world_create(C.byref(cd.world), 4096, -1, C.byref(h)) == 0, A.last_error()
    try:
        for mode in (-1, 0, 1):
            assert lib.vmas_world_set_lidar_compact(h, mode) == 0, A.last_error()
            assert lib.vmas_world_get_lidar_compact(h) == 0  # (no sensors registered)
        for mode in (-2, 2):
            assert lib.vmas_world_set_lidar_compact(h, mode) != 0 and b"mode" in lib.vmas_last_error()
        assert lib.vmas_world_set_lidar_compact(None, 0) != 0 and b"null world" in lib.vmas_last_error()
        assert lib.vmas_world_get_lidar_compact(None) == 0
    finally:
        lib.vmas_world_destroy(h)
