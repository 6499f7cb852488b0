"""Accuracy of the device-side transcendental shortcuts (GPU)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SOFTPLUS, SQRT, DIV, NORM, COS, SIN = range(6)  # include/vmas_debug_hip.h


def _lib():
    from vectorizedmultiagentsimulator_amd import _abi

    lib = _abi.load_library()
    lib.vmas_debug_math.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                    ctypes.c_void_p]
    return lib


def _dev(lib, op, a, b=None):
    ai = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    bi = None if b is None else torch.from_numpy(np.ascontiguousarray(b, np.float32)).cuda()
    out = torch.empty_like(ai)
    assert lib.vmas_debug_math(op, ai.data_ptr(), None if bi is None else bi.data_ptr(), out.data_ptr(), ai.numel(), None) == 0
    return out.cpu().numpy()


def _operands(rng, n):
    """fp32 operands over the magnitudes the step works with (1e-12 .. 1e12, both signs) plus uniform bit patterns of
    normal numbers - the claim under test is 'same bits as the IEEE operation for every operand the exponent scaling
    of the compiler's expansion would not have touched'."""
    mag = np.exp(rng.uniform(np.log(1e-12), np.log(1e12), n)).astype(np.float32)
    sign = np.where(rng.random(n) < 0.5, -1.0, 1.0).astype(np.float32)
    bits = rng.integers(0x10000000, 0x6F000000, n, dtype=np.uint32)  # normal, |x| in ~[2.5e-29, 1.6e29]
    any_normal = bits.view(np.float32) * sign
    return np.concatenate([mag * sign, any_normal]).astype(np.float32)


def test_sqrt_vs_ieee_on_normal_inputs():
    """sqrt_n is the bare v_sqrt_f32 - what hipcc's own sqrtf() is for a normal input (its wrapper only rescales
    denormal inputs; no correction step).  The instruction is a 1-ulp root, NOT a correctly rounded one: against
    numpy's IEEE fp32 sqrt on 10^7 normal inputs (squared lengths 1e-24..1e24 and arbitrary normal bit patterns) it is
    never more than 1 ulp off and exact for ~85 % of them (measured on MI355X: 0.847) - far inside the 1e-5 contract,
    and the reason norms are compared with the reference at 1 ulp, not bitwise."""
    lib = _lib()
    rng = np.random.default_rng(1)
    x = np.abs(_operands(rng, 5_000_000))
    x = np.concatenate([x, np.array([0.0, 1.0, 4.0, 2.0, 1e-30, 3.0e38, np.inf], np.float32)])
    got = _dev(lib, SQRT, x)
    want = np.sqrt(x)
    fin = np.isfinite(want)
    ulps = np.abs(got[fin].view(np.int32).astype(np.int64) - want[fin].view(np.int32).astype(np.int64))
    exact = float((ulps == 0).mean())
    print(f"sqrt_n: max {int(ulps.max())} ulp from the correctly rounded root, {exact:.6f} of {x.size} inputs exact")
    assert ulps.max() <= 1 and exact > 0.8
    assert np.array_equal(got[~fin], want[~fin])
    for v in (0.0, 1.0, 4.0):
        assert _dev(lib, SQRT, np.array([v], np.float32))[0] == np.float32(np.sqrt(v))


def test_division_is_ieee_wherever_no_exponent_scaling_is_needed():
    """a / rcp_of(b) runs hipcc's division sequence without v_div_scale / v_div_fmas: IEEE quotient bit for bit on 10^7
    operand pairs whose quotient and reciprocal stay inside the normal range; 0, inf and NaN operands keep their IEEE
    results through v_div_fixup."""
    lib = _lib()
    rng = np.random.default_rng(2)
    a, b = _operands(rng, 5_000_000), _operands(rng, 5_000_000)
    with np.errstate(over="ignore", under="ignore", divide="ignore", invalid="ignore"):
        q = a.astype(np.float64) / b.astype(np.float64)
    ok = (np.abs(q) > 1e-30) & (np.abs(q) < 1e30) & (np.abs(b) > 1e-30) & (np.abs(b) < 1e30)
    a, b = a[ok], b[ok]
    got = _dev(lib, DIV, a, b)
    want = (a / b).astype(np.float32)
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert a.size > 6_000_000
    assert not bad.any(), f"{int(bad.sum())} of {a.size} quotients differ from IEEE, first {a[bad][0]!r}/{b[bad][0]!r}: {got[bad][0]!r} vs {want[bad][0]!r}"
    sa = np.array([0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.0, 5.0, np.inf, 1.0], np.float32)
    sb = np.array([1.0, 0.0, 0.0, 1.0, 2.0, 1.0, 0.0, np.inf, np.inf, np.nan], np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = sa / sb
    got = _dev(lib, DIV, sa, sb)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)])


def test_norm_matches_torch_cpu_vector_norm():
    """norm2(x, y) = sqrt(fma(y, y, x*x)): the radicand is bitwise torch's (CPU linalg.vector_norm over a size-2 dim,
    probed in SURVEY.md); the root is v_sqrt_f32, a 1-ulp instruction (test above) - so the norm is within 1 ulp of the
    reference's on every one of 2 * 10^6 vectors and bitwise equal on ~85 % of them."""
    lib = _lib()
    rng = np.random.default_rng(3)
    v = torch.from_numpy((rng.standard_normal((2_000_000, 2)) * np.exp(rng.uniform(-8, 8, (2_000_000, 1)))).astype(np.float32))
    want = torch.linalg.vector_norm(v, dim=-1).numpy()
    got = _dev(lib, NORM, v[:, 0].numpy(), v[:, 1].numpy())
    ulps = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    exact = float((ulps == 0).mean())
    print(f"norm2: max {int(ulps.max())} ulp from torch's CPU vector_norm, {exact:.6f} bitwise equal")
    assert ulps.max() <= 1 and exact > 0.8


def test_sincos_accuracy_vs_float64():
    """cos/sin of the entity rotations (write_trig, ocml sincosf): <= 2 ulp against float64 on |x| <= 100 rad (measured
    maximum on MI355X: 1.52 ulp)."""
    lib = _lib()
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.uniform(-7, 7, 1 << 20), rng.uniform(-100, 100, 1 << 20),
                        np.array([0.0, np.pi / 2, np.pi, -np.pi / 2, 1e-8], np.float64)]).astype(np.float32)
    for op, fn in ((COS, np.cos), (SIN, np.sin)):
        got = _dev(lib, op, x).astype(np.float64)
        ref = fn(x.astype(np.float64))
        err = np.abs(got - ref) / np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        print(f"sincosf op {op}: max {err.max():.2f} ulp")
        assert err.max() <= 2.0, f"op {op}: {err.max():.2f} ulp at x={x[err.argmax()]!r}"


def test_softplus_accuracy_vs_float64():
    """softplus0 = max(0,x) + log1p_unit(exp(-|x|)) replaces ocml's log1pf by a 20-instruction
    formulation; it must stay in the accuracy class of the libms it is compared with
    (torch/SLEEF and glibc are <= 1 ulp on their pieces): <= 2 ulp against float64."""
    lib = _lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-30, 30, 1 << 20), rng.uniform(-1, 1, 1 << 18), rng.uniform(-100, 100, 1 << 18),
                        np.array([0.0, -0.0, 1e-8, -1e-8, 88.0, -88.0, 200.0, -200.0])]).astype(np.float32)
    got = _dev(lib, SOFTPLUS, x).astype(np.float64)
    x64 = x.astype(np.float64)
    ref = np.maximum(x64, 0) + np.log1p(np.exp(-np.abs(x64)))
    ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
    err = np.abs(got - ref) / ulp
    # x >= 0 is the only branch whose value survives: for x < 0 (dist beyond dist_min) the
    # reference zeroes the force (core.py:2834-2838), so only a loose sanity bound there
    pos = x >= 0
    assert err[pos].max() <= 2.5, f"max error {err[pos].max():.2f} ulp at x={x[pos][err[pos].argmax()]}"
    assert err[~pos].max() <= 4.0, f"max error {err[~pos].max():.2f} ulp at x={x[~pos][err[~pos].argmax()]}"
    print(f"softplus0: x>=0 max {err[pos].max():.2f} ulp (mean {err[pos].mean():.3f}); x<0 max {err[~pos].max():.2f} ulp")
