"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the restatement is the checker
of everything else, so it gets checked for out-of-bounds indexing and undefined arithmetic itself): every entry point on
fixtures that cover all pair types, joints, per-environment inputs, LIDAR and the geometric queries."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

SCRIPT = r'''
import sys, numpy as np
sys.path[:0] = [ROOT, ROOT + "/tests"]
from golden_util import load
from oracle.oracle import Oracle
n = 0
for name in ("balance_n4", "transport_2pkg", "navigation_n8", "football_5v5", "waterfall", "pollock", "soup_solid",
             "soup_hollow", "wind_flocking", "joint_passage", "band_4env"):
    g = load(name)
    o = Oracle(g.spec)
    for t in range(min(g.T, 3)):
        st, ft = np.ascontiguousarray(g.state0[t]).copy(), np.ascontiguousarray(g.ft_in[t]).copy()
        jfr = None if g.jfr is None else np.ascontiguousarray(g.jfr[t])
        eg = None if g.egrav is None else np.ascontiguousarray(g.egrav[t])
        o.pair_mask(st)
        if g.spec.lidars:
            o.cast_rays(st)
        if g.queries:
            o.queries(st, g.queries)
        o.step(st, ft, joint_fixed_rot=jfr, entity_gravity=eg, pair_mask=np.ascontiguousarray(g.masks[t, 0]), first_substep=0, n_substeps=1)
        o.step_exact(st, ft, joint_fixed_rot=jfr, entity_gravity=eg)
        o.step(st, ft, joint_fixed_rot=jfr, entity_gravity=eg, threads=3)
        n += 1
print("sanitized steps:", n)
'''


@pytest.mark.skipif(not os.path.exists(subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
                                       .stdout.strip()), reason="gcc's libasan is not installed")
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    lib = str(tmp_path / "libvmas_oracle_san.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", lib,
                           os.path.join(ROOT, "oracle", "vmas_oracle.c"), "-lm"])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, VMAS_ORACLE_LIB=lib, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([sys.executable, "-c", SCRIPT.replace("ROOT", repr(ROOT))], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    assert "sanitized steps:" in out.stdout
