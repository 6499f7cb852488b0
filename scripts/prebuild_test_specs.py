"""Compile, ahead of time and without a GPU, the run-time specialisations the GPU tests ask for (tests/test_specialize_gpu.py)
into the package's on-disk cache, so that the test box loads them instead of compiling (~10-60 s each; eight at a time here):
python scripts/prebuild_test_specs.py [--quiet]   (also run by __graft_entry__.build())"""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load  # noqa: E402
from vectorizedmultiagentsimulator_amd import _abi as A  # noqa: E402
from vectorizedmultiagentsimulator_amd import specialize as S  # noqa: E402

from golden_util import FIXTURES  # noqa: E402

lib = A.load_library()
BATCH = {"balance_n3": 4096, "transport_2pkg": 1024, "all_joint_passage_size": 700, "ball_trajectory": 1000, "give_way": 4096,
         "all_wheel": 64 * 7 + 3}  # (tests/test_specialize_gpu.py; every other fixture: 640 + 7 environments)


def plan(name):
    """(label, source or None, note): the world is planned here, on the calling thread."""
    B = BATCH.get(name, 647)
    g = load(name)
    cd = g.spec.to_ctypes()
    h = C.c_void_p()
    assert lib.vmas_world_create(C.byref(cd.world), B, -1, C.byref(h)) == 0, A.last_error()
    try:
        meta, words = S.schedule(h)
    finally:
        lib.vmas_world_destroy(h)
    if meta[23] >= 0:
        return f"{name}", None, "has a built-in specialisation"
    try:
        return f"{name} {B}", S.render(meta, words, int(g.spec.substeps), 0), ""
    except S.SpecializeError as e:
        return f"{name} {B}", None, f"refused: {str(e)[:100]}"


if __name__ == "__main__":
    from concurrent.futures import ThreadPoolExecutor  # (hipcc runs as a subprocess: threads are enough)

    quiet = "--quiet" in sys.argv
    jobs = [plan(name) for name in FIXTURES]  # planned one after the other, compiled eight at a time
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        paths = list(pool.map(lambda j: S.code_object(j[1]) if j[1] is not None else None, jobs))
    if not quiet:
        for (label, src, note), p in zip(jobs, paths):
            print(label, note if p is None else f"-> {os.path.basename(p)} {os.path.getsize(p)}", flush=True)
