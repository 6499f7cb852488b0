"""The output-set pool of the fused post-steps (fused._Post): ``Environment.step`` hands out tensors that look fresh on every
step - as the reference's do - while the sets behind them are recycled as soon as the caller holds nothing of them.  The
rule that makes it safe is checked here on CPU tensors: a set is reused ONLY when neither one of its handed-out tensor
objects nor any view onto its storage is alive outside the set."""
import types

import pytest
import torch

from vectorizedmultiagentsimulator_amd import _abi as A
from vectorizedmultiagentsimulator_amd import fused as F


class _DummyPost(F._Post):
    D = 5

    def _new_set(self):
        n, B = self.n, self.B
        st = F._OutSet()
        st.base, v = self._carve([("obs", (n, B, self.D), torch.float32), ("rew", (n, B), torch.float32), ("done", (B,), torch.bool)])
        st.v = v
        st.obs, st.rew, st.done = list(v["obs"].unbind(0)), list(v["rew"].unbind(0)), v["done"]
        st.extra, st.infos = None, {"x": st.rew[0]}
        st.tensors = tuple(st.obs) + tuple(st.rew) + (st.done,)
        st.buffers = A.TransportBuffers()
        return st

    def prepare(self, dedicated=False):
        st = self._acquire(dedicated)
        return st, (list(st.obs), list(st.rew), st.done, [dict(st.infos) for _ in range(self.n)])


@pytest.fixture()
def post():
    env = types.SimpleNamespace(num_envs=7, device=torch.device("cpu"), agents=[object(), object(), object()],
                                steps=torch.zeros(7), max_steps=None)
    return _DummyPost(env)


def test_views_are_aligned_disjoint_and_typed(post):
    st, (obs, rew, done, infos) = post.prepare()
    assert obs[0].shape == (7, 5) and rew[0].shape == (7,) and done.dtype == torch.bool and done.shape == (7,)
    ptrs = sorted((t.data_ptr(), t.numel() * t.element_size()) for t in list(obs) + list(rew) + [done])
    assert all(a + n <= b for (a, n), (b, _) in zip(ptrs, ptrs[1:])), "output views overlap"
    base = st.base.data_ptr()  # (the GPU allocator's blocks are 512-byte aligned; the views keep 256 bytes relative to it)
    assert all((st.v[k].data_ptr() - base) % 256 == 0 for k in ("obs", "rew", "done"))
    st.base.zero_()
    obs[1].fill_(3.0)
    assert float(st.v["obs"].sum()) == 3.0 * 35 and float(st.v["rew"].sum()) == 0.0 and not bool(done.any())


def test_a_set_is_recycled_only_when_the_caller_holds_nothing_of_it(post):
    if F._use_count is None:
        pytest.skip("this torch build has no storage use count: every step allocates, as before")
    a, ra = post.prepare()
    b, rb = post.prepare()
    assert a is not b, "the first set is still held by the caller (ra)"
    del rb
    c, rc = post.prepare()
    assert c is b, "a set the caller has dropped entirely is reused"
    del rc
    # holding ONE of the tensor objects keeps the set busy ...
    keep = ra[0][1]
    del ra
    assert not a.free()
    del keep
    assert a.free()
    # ... so does a view the caller made of one (its own tensor object, the set's storage) ...
    st, res = post.prepare()
    view = res[0][0][:2]
    which = st
    del res
    assert not which.free()
    del view
    assert which.free()
    # ... and an info dictionary, or the done tensor
    st, res = post.prepare()
    info = res[3][2]
    del res
    assert not st.free()
    del info
    assert st.free()


def test_steady_state_uses_two_sets(post):
    if F._use_count is None:
        pytest.skip("no storage use count")
    seen = set()
    res = None
    for _ in range(50):  # the usual loop: the previous step's results are still bound while the next step is made
        st, res = post.prepare()
        seen.add(id(st))
    assert len(seen) == 2 and len(post._pool) == 2


def test_hoarding_caller_gets_fresh_sets_and_the_pool_stays_bounded(post):
    if F._use_count is None:
        pytest.skip("no storage use count")
    hoard = [post.prepare()[1] for _ in range(3 * post.POOL_MAX_SETS)]
    ptrs = {r[0][0].data_ptr() for r in hoard}
    assert len(ptrs) == len(hoard), "a set still held by the caller was handed out again"
    assert len(post._pool) == post.POOL_MAX_SETS
    del hoard
    assert all(s.free() for s in post._pool)


def test_static_and_dedicated_sets(post):
    d1 = post.prepare(dedicated=True)[0]
    d2 = post.prepare(dedicated=True)[0]
    assert d1 is not d2 and not post._pool
    post.static_outputs = True
    s1, r1 = post.prepare()
    s2, r2 = post.prepare()
    assert s1 is s2 and not post._pool  # (HIP-graph replay: the captured pointers must not change)


def test_an_autograd_graph_over_an_observation_keeps_its_set_busy(post):
    """``policy(obs[i])`` saves the observation for backward() - a C++ handle on the SAME TensorImpl.  While the
    graph is alive the set must not be written again, on any torch build: the TensorImpl use count is compared as well as
    the Python reference count."""
    if not F.POOLING:
        pytest.skip("output pooling is off on this torch build: every step allocates")
    w = torch.ones(5, requires_grad=True)
    st, res = post.prepare()
    res[0][1].fill_(2.0)
    loss = (res[0][1] * w).sum()  # the graph saves obs[1]
    del res
    assert not st.free(), "an observation saved by an autograd graph must keep its set out of the pool"
    for _ in range(6):  # later steps: other sets
        st2, r2 = post.prepare()
        assert st2 is not st
        r2[0][1].fill_(7.0)
        del r2
    loss.backward()
    assert torch.equal(w.grad, torch.full((5,), 14.0)), "backward() must see the observation as it was handed out"
    del loss
    assert st.free()


def test_impl_use_count_is_a_witness_of_its_own(post):
    """The TensorImpl count alone (Python counts equal) marks a set busy."""
    if not F.POOLING:
        pytest.skip("output pooling is off")
    st, res = post.prepare()
    del res
    assert st.free()
    st.ic0 = [c - 1 for c in st.ic0]  # as if a C++ handle existed that the Python count did not show
    assert not st.free()


def test_pooling_can_be_switched_off(post, monkeypatch):
    monkeypatch.setattr(F, "POOLING", False)
    a, ra = post.prepare()
    del ra
    b, rb = post.prepare()
    assert a is not b and not post._pool, "with pooling off every step gets a set of its own"
