"""CPU-side checks of the drop-in boundary: libvmas_hip.so loads without a GPU and
exports every function include/*.h declares; struct layouts agree with ctypes."""
import ctypes
import os
import re
import subprocess

import pytest

from vectorizedmultiagentsimulator_amd import _abi

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("vmas_hip.h", "vmas_env_hip.h", "vmas_debug_hip.h")]


def _declared_functions():
    src = "".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vmas_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_abi.LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "vectorizedmultiagentsimulator_amd", "csrc", "build.sh")])
    return _abi.load_library()


def test_header_and_loader_agree():
    assert _declared_functions() == sorted(_abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_functions():
        assert hasattr(lib, name), f"libvmas_hip.so does not export {name}"
    assert lib.vmas_abi_version() == _abi.ABI_VERSION


def test_struct_sizes_match_the_c_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof."""
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "vmas_hip.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(VmasEntityDesc), sizeof(VmasPairDesc),"
        " sizeof(VmasJointDesc), sizeof(VmasWorldDesc), sizeof(VmasStepArgs), sizeof(VmasLidarDesc),"
        " offsetof(VmasWorldDesc, entities), offsetof(VmasLidarDesc, targets));return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [
        ctypes.sizeof(_abi.EntityDesc), ctypes.sizeof(_abi.PairDesc), ctypes.sizeof(_abi.JointDesc),
        ctypes.sizeof(_abi.WorldDesc), ctypes.sizeof(_abi.StepArgs), ctypes.sizeof(_abi.LidarDesc),
        _abi.WorldDesc.entities.offset, _abi.LidarDesc.targets.offset,
    ]
    assert got == want


def test_env_struct_sizes_match_the_c_header(tmp_path):
    """Same for include/vmas_env_hip.h (structs passed by pointer AND copied into kernel arguments)."""
    names = ["VmasSpawnOp", "VmasResetTerm", "VmasResetArgs", "VmasAgentScript", "VmasFootballDesc", "VmasFootballBuffers", "VmasActionSlot", "VmasIngestArgs", "VmasStepLimit", "VmasBalanceDesc", "VmasBalanceBuffers",
             "VmasTransportDesc", "VmasTransportBuffers", "VmasNavigationDesc", "VmasNavigationBuffers"]
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "vmas_env_hip.h"\nint main(){'
        + "".join(f'printf("%zu ", sizeof({n}));' for n in names)
        + 'printf("%zu %zu %zu\\n", offsetof(VmasNavigationBuffers, limit), offsetof(VmasNavigationDesc, agent_radius),'
          " offsetof(VmasIngestArgs, agents));return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    py = [_abi.SpawnOp, _abi.ResetTerm, _abi.ResetArgs, _abi.AgentScript, _abi.FootballDesc, _abi.FootballBuffers, _abi.ActionSlot, _abi.IngestArgs, _abi.StepLimit, _abi.BalanceDesc, _abi.BalanceBuffers, _abi.TransportDesc,
          _abi.TransportBuffers, _abi.NavigationDesc, _abi.NavigationBuffers]
    want = [ctypes.sizeof(t) for t in py] + [_abi.NavigationBuffers.limit.offset, _abi.NavigationDesc.agent_radius.offset,
                                             _abi.IngestArgs.agents.offset]
    assert got == want


def test_errors_are_reported_not_thrown(lib):
    """Error convention of the ABI: negative return code + vmas_last_error()."""
    import torch

    from golden_util import load

    g = load("balance_n3")
    cd = g.spec.to_ctypes()
    h = ctypes.c_void_p()
    assert lib.vmas_world_create(None, 4, 0, ctypes.byref(h)) < 0
    assert b"null" in lib.vmas_last_error()
    assert lib.vmas_world_create(ctypes.byref(cd.world), 0, 0, ctypes.byref(h)) < 0
    assert b"batch" in lib.vmas_last_error()
    cd.world.abi_version = 99
    assert lib.vmas_world_create(ctypes.byref(cd.world), 4, 0, ctypes.byref(h)) < 0
    assert b"ABI" in lib.vmas_last_error()
    cd.world.abi_version = _abi.ABI_VERSION
    if not torch.cuda.is_available():
        # no GPU in the build container: creation must fail loudly, not fall back
        assert lib.vmas_world_create(ctypes.byref(cd.world), 4, 0, ctypes.byref(h)) < 0
        assert lib.vmas_last_error()


def test_product_has_no_cpu_fallback():
    """World.step() on a CPU device raises; nothing under the package imports oracle/."""
    import torch

    from vectorizedmultiagentsimulator_amd.backend import VmasHipError
    from vectorizedmultiagentsimulator_amd.scenarios.balance import Scenario

    sc = Scenario()
    w = sc.env_make_world(4, "cpu", n_agents=3)
    sc.env_reset_world_at(None)
    with pytest.raises(VmasHipError):
        w.step()
    pkg = os.path.join(ROOT, "vectorizedmultiagentsimulator_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "vmas_oracle" not in txt, f


def test_env_entry_points_report_errors_not_throw(lib):
    """include/vmas_env_hip.h: malformed arguments come back as -1 + message, without touching a GPU."""
    import ctypes as C

    args = _abi.IngestArgs()
    args.n_agents = _abi.ENV_MAX_AGENTS + 1
    assert lib.vmas_env_ingest_actions(C.byref(args), 8, None, C.c_void_p(64), 64, None, None) == -1
    assert b"n_agents" in lib.vmas_last_error()
    args.n_agents = 1  # slot 0 has no action pointer
    assert lib.vmas_env_ingest_actions(C.byref(args), 8, None, C.c_void_p(64), 64, None, None) == -1
    assert b"malformed action slot" in lib.vmas_last_error()
    d, b = _abi.BalanceDesc(), _abi.BalanceBuffers()
    assert lib.vmas_balance_post_step(C.byref(d), C.byref(b), 8, C.c_void_p(64), 64, None) == -1  # n_agents = 0
    d.n_agents = 2
    assert lib.vmas_balance_post_step(C.byref(d), C.byref(b), 8, C.c_void_p(64), 4, None) == -1  # ld < batch
    assert b"ld" in lib.vmas_last_error()
    assert lib.vmas_balance_post_step(C.byref(d), C.byref(b), 8, C.c_void_p(64), 64, None) == -1  # null buffers
    assert b"null buffer" in lib.vmas_last_error()
    t, tb = _abi.TransportDesc(), _abi.TransportBuffers()
    t.n_agents, t.n_packages = 2, _abi.ENV_MAX_PACKAGES + 1
    assert lib.vmas_transport_post_step(C.byref(t), C.byref(tb), 8, C.c_void_p(64), 64, None) == -1
    n, nb = _abi.NavigationDesc(), _abi.NavigationBuffers()
    assert lib.vmas_navigation_post_step(C.byref(n), C.byref(nb), 8, C.c_void_p(64), 64, None) == -1
    f, fb = _abi.FootballDesc(), _abi.FootballBuffers()
    assert lib.vmas_football_post_step(C.byref(f), C.byref(fb), 8, C.c_void_p(64), 64, None) == -1
    assert lib.vmas_world_step_env(None, None, None, 64, None, None, None, 1, None, None, None) == -1
    assert lib.vmas_world_reserve_epilogue(None, 1, 0) == -1
    assert b"null world" in lib.vmas_last_error()
    # round 5: the validation entries and the gated step
    assert lib.vmas_world_step_env_gated(None, None, None, 64, None, None, None, 1, None, None, None) == -1
    good = _abi.IngestArgs()
    assert lib.vmas_env_validate_actions(C.byref(good), 8, None, C.c_void_p(64), 64, None, None, None) == -1
    assert b"flag word" in lib.vmas_last_error()
    assert lib.vmas_env_validate_begin(C.byref(good), 8, None, C.c_void_p(64), 64, None, None, None) == -1
    assert lib.vmas_env_validate_end(None, 1, None) == -1
    assert lib.vmas_host_word_gate(None) is None
    lib.vmas_host_word_destroy(None)  # (a no-op, like free(NULL))
