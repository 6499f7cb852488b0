"""Run-time specialisation (vectorizedmultiagentsimulator_amd/specialize.py), host side: the schedule of a world the
library has no built-in specialisation for is rendered as the tables of csrc/vmas_spec_kernel.h and compiled for gfx950
(hipcc cross-compiles without a GPU); the GPU tests load it and compare it with the interpreter bit for bit."""
import ctypes as C
import importlib
import os
import subprocess

import pytest

from vectorizedmultiagentsimulator_amd import _abi as A
from vectorizedmultiagentsimulator_amd import specialize as S


def _planned(name, batch, **kw):
    lib = A.load_library()
    sc = importlib.import_module(f"vectorizedmultiagentsimulator_amd.scenarios.{name}").Scenario()
    world = sc.env_make_world(batch, "cpu", **kw)
    cd = world.spec.to_ctypes()
    h = C.c_void_p()
    assert lib.vmas_world_create(C.byref(cd.world), batch, -1, C.byref(h)) == 0, A.last_error()
    try:
        hint = getattr(world, "epilogue_hint", None)
        if hint is not None:
            assert lib.vmas_world_reserve_epilogue(h, *hint) == 0
        meta, words = S.schedule(h)
    finally:
        lib.vmas_world_destroy(h)
    return world, meta, words


def test_render_and_compile_a_world_without_a_builtin_specialisation(tmp_path):
    world, meta, words = _planned("balance", 32768, n_agents=3)  # BASELINE config 1's world at config 2's batch
    assert meta[23] == -1, "balance n_agents=3 has no built-in specialisation"
    src = S.render(meta, words, int(world.spec.substeps), 1)
    assert "struct SpecRT" in src and "vmas_rt_lean_t0" in src and "vmas_rt_multi_e1_o1_t0" in src and "vmas_rt_check" in src
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    path = S.code_object(src, cache_dir=str(tmp_path))
    assert os.path.getsize(path) > 10000
    again = S.code_object(src, cache_dir=str(tmp_path))  # the cache: same key, no second compilation
    assert again == path
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm"
    if os.path.exists(nm):
        syms = subprocess.run([nm, path], capture_output=True, text=True).stdout
        for want in ("vmas_rt_lean_t0", "vmas_rt_lean_t1", "vmas_rt_multi_e0_o0_t0", "vmas_rt_multi_e3_o1_t1", "vmas_rt_multi_e1_o1_t0",
                     "vmas_rt_check"):
            assert want in syms, want


def test_builtin_worlds_are_not_recompiled():
    _, meta, _ = _planned("balance", 32768, n_agents=4)
    assert meta[23] >= 0  # SpecBalance4 serves it: specialize() returns at once


def test_worlds_whose_items_do_not_fit_lds_are_refused():
    meta = [8, 2, 1, 4, 0, 0, 0, 0, 0, 1, 1, 10, 0, 0, 3, 1, 0, 0, 0, 0, 0, 0, 1, -1]  # items_in_lds == 0
    with pytest.raises(S.SpecializeError):
        S.render(meta, [0, 0, 0, 0], 1, 0)


@pytest.mark.parametrize("name,kw,batches", [
    ("balance", dict(n_agents=4), (32768, 65536, 131072, 262144, 1048576)),   # SpecBalance4 / SpecBalance4Wide
    ("transport", {}, (16384, 131072, 1048576)),                              # SpecTransport
    ("navigation", dict(n_agents=8), (4096, 8192, 16384, 65536, 131072)),     # SpecNavigation8Shard / SpecNavigation8
])
def test_the_planner_lands_on_a_geometry_with_a_builtin_specialisation(name, kw, batches):
    """The BASELINE worlds at their BASELINE batch sizes and beyond must be planned onto a geometry one of the generated
    specialisations serves (a planning world: the decision is the host's).  Round 3 found balance at 1 M environments on the
    interpreter - 179 us instead of 99 - because two geometries tied and the tie went to the one without tables."""
    for B in batches:
        _, meta, _ = _planned(name, B, **kw)
        assert meta[23] >= 0, f"{name} at {B} environments is planned at {meta[0]} waves per tile (sharing mode {meta[1]}): no built-in specialisation serves that geometry"


def _race_worker(src, cache_dir, fake, q):
    from vectorizedmultiagentsimulator_amd import specialize as S2

    try:
        q.put(("ok", S2.code_object(src, cache_dir=cache_dir, hipcc=fake)))
    except Exception as e:  # noqa: BLE001
        q.put(("err", repr(e)))


def test_ranks_racing_on_a_cold_cache_compile_once_and_all_get_the_file(tmp_path):
    """One rank per GPU, each `make_env(..., specialize=True)` on a cold cache: every process must come back with the same
    whole file, and the lock must keep all but one from compiling (round 3 staged every compile under one shared temporary
    name: a second rank's rename could find it gone, or a half-written file could pass the size check)."""
    import multiprocessing as mp
    import stat

    fake = tmp_path / "fake_hipcc"
    log = tmp_path / "invocations.log"
    # a stand-in compiler: slow enough for the ranks to overlap, writes its output in two halves
    fake.write_text(f"""#!/usr/bin/env python3
import sys, time
out = sys.argv[sys.argv.index('-o') + 1]
open({str(log)!r}, 'a').write('x\\n')
with open(out, 'wb') as f:
    f.write(b'A' * 5000); f.flush(); time.sleep(0.5); f.write(b'B' * 5000)
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    cache = tmp_path / "cache"
    src = "// not a real kernel: the stand-in compiler does not read it\n"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_race_worker, args=(src, str(cache), str(fake), q)) for _ in range(4)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=120) for _ in procs]
    for p_ in procs:
        p_.join(timeout=60)
    assert all(kind == "ok" for kind, _ in res), res
    paths = {p_ for _, p_ in res}
    assert len(paths) == 1
    path = paths.pop()
    assert open(path, "rb").read() == b"A" * 5000 + b"B" * 5000, "a reader saw a half-written code object"
    assert log.read_text().count("x") == 1, "the lock should have let ONE process compile"
    assert sorted(os.listdir(cache)) == [os.path.basename(path)], "temporary / lock files left behind"
