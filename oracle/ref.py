"""The REAL reference (VMAS, /root/reference) as a checker and CPU baseline - TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__`` (build) and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package never does (tests/test_abi_symbols.py greps for it).

``/root/reference`` exists in the build container only.  ``build_ref()`` (run by ``__graft_entry__.build()`` there)
byte-compiles the reference package and its tests FROM THE SOURCES WHERE THEY LIE into ``oracle/_ref/`` - sourceless
``.pyc`` files, i.e. build outputs like ``oracle/_ref``'s C counterparts would be: git-ignored (never in history), not
gpurun-ignored (they travel to the GPU box with the built ``.so`` files).  No reference source text is copied.

``import_vmas()`` returns the reference package: from ``/root/reference`` when it is there, else from ``oracle/_ref``;
``gym`` (absent from the image) is the constructor-only stand-in in ``oracle/ref_shim``.
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.util
import json
import os
import pathlib
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("VMAS_REFERENCE_SRC", "/root/reference")  # (the override lets the CPU suite exercise the sourceless tree)
REF = os.path.join(HERE, "_ref")
SHIM = os.path.join(HERE, "ref_shim")
_STAMP = os.path.join(REF, "STAMP")
_MANIFEST = os.path.join(REF, "MANIFEST.json")


def _magic() -> str:
    return importlib.util.MAGIC_NUMBER.hex()


def build_ref(force: bool = False) -> bool:
    """Byte-compile /root/reference/{vmas,tests} into oracle/_ref (sourceless layout: pkg/mod.pyc).  Returns False when
    the reference is not present (GPU box: the prebuilt files are used as they are)."""
    if not os.path.isdir(os.path.join(SRC, "vmas")):
        return False
    if os.path.exists(_STAMP) and not force and open(_STAMP).read().split()[0] == _magic():
        return True
    shutil.rmtree(REF, ignore_errors=True)
    n = 0
    for top in ("vmas", "tests"):
        for dirpath, dirnames, filenames in os.walk(os.path.join(SRC, top)):
            dirnames[:] = [d for d in dirnames if d != "__pycache__"]
            rel = os.path.relpath(dirpath, SRC)
            for fn in filenames:
                if not fn.endswith(".py"):
                    continue
                dst = os.path.join(REF, rel, fn + "c")
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                py_compile.compile(os.path.join(dirpath, fn), cfile=dst, dfile=os.path.join("reference", rel, fn), doraise=True,
                                   invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
                n += 1
    # data files the package reads at run time (road_traffic's map, vmas/scenarios_data/**): copied next to the byte code -
    # into the git-ignored build output only, like the .pyc files
    data_src = os.path.join(SRC, "vmas", "scenarios_data")
    if os.path.isdir(data_src):
        shutil.copytree(data_src, os.path.join(REF, "vmas", "scenarios_data"), dirs_exist_ok=True)
    # what the reference's own discovery (tests/test_vmas.py::scenario_names, vmas/scenarios/__init__.py::load) finds by
    # globbing ``vmas/scenarios/**/*.py``: kept as a manifest, because a sourceless tree has no ``*.py`` to glob
    names = sorted(os.path.splitext(fn)[0] for _, _, fns in os.walk(os.path.join(SRC, "vmas", "scenarios")) for fn in fns
                   if fn.endswith(".py") and not fn.startswith("__"))
    with open(_MANIFEST, "w") as f:
        json.dump({"scenarios": names}, f)
    with open(_STAMP, "w") as f:
        f.write(f"{_magic()} {n} modules byte-compiled from {SRC} (VMAS reference, sourceless)\n")
    return True


def scenario_manifest():
    """Scenario names the reference ships (from the sources when they are here, else the manifest build_ref() wrote)."""
    if os.path.isdir(os.path.join(SRC, "vmas")):
        return sorted(os.path.splitext(fn)[0] for _, _, fns in os.walk(os.path.join(SRC, "vmas", "scenarios")) for fn in fns
                      if fn.endswith(".py") and not fn.startswith("__"))
    with open(_MANIFEST) as f:
        return json.load(f)["scenarios"]


@contextlib.contextmanager
def _sourceless_glob():
    """The reference discovers its scenarios by globbing ``*.py`` (tests/test_vmas.py:18-24).  On the byte-compiled tree
    those files are ``*.pyc`` (same stems): while a reference module is being executed from oracle/_ref, a ``*.py`` glob
    INSIDE that tree also yields the ``*.pyc`` files.  Nothing outside oracle/_ref is affected."""
    if root() != REF:
        yield
        return
    orig = pathlib.Path.glob

    def glob(self, pattern, *a, **kw):
        hits = list(orig(self, pattern, *a, **kw))
        if isinstance(pattern, str) and pattern.endswith(".py") and os.path.abspath(str(self)).startswith(REF):
            hits += list(orig(self, pattern + "c", *a, **kw))
        return iter(hits)

    pathlib.Path.glob = glob
    try:
        yield
    finally:
        pathlib.Path.glob = orig


def available() -> bool:
    return os.path.isdir(os.path.join(SRC, "vmas")) or (os.path.exists(_STAMP) and open(_STAMP).read().split()[0] == _magic())


def root() -> str:
    """Directory that holds the importable ``vmas`` package and its ``tests``."""
    return SRC if os.path.isdir(os.path.join(SRC, "vmas")) else REF


def import_vmas():
    if not available():
        raise ImportError("the reference is neither at /root/reference nor byte-compiled under oracle/_ref "
                          "(run __graft_entry__.build() in the build container)")
    for p in (root(), SHIM):
        if p not in sys.path:
            sys.path.insert(0, p)
    return importlib.import_module("vmas")


def scenario_class(name: str):
    """The reference's Scenario class by name (``vmas.scenarios.load`` walks ``*.py`` files, which a sourceless tree
    does not have: import the module instead - same class either way)."""
    import_vmas()
    name = name[:-3] if name.endswith(".py") else name
    for pkg in ("vmas.scenarios", "vmas.scenarios.debug", "vmas.scenarios.mpe"):
        try:
            return importlib.import_module(f"{pkg}.{name}").Scenario
        except ModuleNotFoundError:
            continue
    raise ValueError(f"reference scenario {name!r} not found")


def make_env(scenario, **kw):
    """``vmas.make_env`` of the reference, scenario names resolved by import (works on the sourceless tree too)."""
    vmas = import_vmas()
    if isinstance(scenario, str):
        scenario = scenario_class(scenario)()
    return vmas.make_env(scenario=scenario, **kw)


def load_test_module(relpath: str):
    """A module of the reference's own test-suite, e.g. ``tests/test_scenarios/test_balance.py``."""
    import_vmas()
    base = os.path.join(root(), relpath)
    path = base if os.path.exists(base) else base + "c"
    name = "reftest_" + relpath.replace("/", "_").rsplit(".", 1)[0]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    with _sourceless_glob():  # (parametrize marks over scenario_names() are evaluated while the module executes)
        spec.loader.exec_module(mod)
    return mod
