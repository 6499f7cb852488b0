"""Where do the compacted kernel and the interpreter part ways on a held action, and which of them agrees with the oracle?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vectorizedmultiagentsimulator_amd.environment import make_env
from oracle.oracle import Oracle

kw = dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
envs = [make_env("football", num_envs=B, device="cuda:0", seed=2, validate_actions=False, **kw) for _ in range(2)]
acts = [envs[0].get_random_action(a) for a in envs[0].agents]
for env in envs:
    for _ in range(20):
        env.step([a.clone() for a in acts])
cp, it = (e.world._get_backend() for e in envs)
cp.set_compact(1); it.set_compact(0)
sa, sb = envs[0].world._state, envs[1].world._state
assert torch.equal(sa.view(torch.int32), sb.view(torch.int32))
spec = envs[0].world.spec
o = Oracle(spec)
nE, nA = spec.n_entities, spec.n_agents
for t in range(1500):
    prev = sa.clone(); ft_prev = envs[0].world._agent_ft.clone()
    cp.step(); it.step()
    if not torch.equal(sa.view(torch.int32), sb.view(torch.int32)):
        d = (sa != sb) & ~(torch.isnan(sa) & torch.isnan(sb))
        bits = sa.view(torch.int32) != sb.view(torch.int32)
        envs_bad = torch.nonzero(bits.any(0).any(0)).flatten()
        print("first divergence at held step", t, ": values differing", int(bits.sum()), "value-differing (non-NaN-pair)", int(d.sum()),
              "environments", envs_bad.numel(), "nan in compact", int(torch.isnan(sa).sum()), "nan in interp", int(torch.isnan(sb).sum()))
        e = int(envs_bad[0])
        idx = torch.nonzero(bits[:, :, e])
        for (i, f) in idx[:6].tolist():
            print("  env", e, "entity", i, "field", f, "compact %.9g interp %.9g prev %.9g" % (float(sa[i, f, e]), float(sb[i, f, e]), float(prev[i, f, e])))
        # the oracle on this environment's previous state
        ld = 64
        st = np.zeros((nE, 6, ld), np.float32); ft = np.zeros((nA, 3, ld), np.float32)
        st[:, :, 0] = prev[:nE, :, e].cpu().numpy(); ft[:, :, 0] = ft_prev[:nA, :, e].cpu().numpy()
        o.step(st, ft, batch=1)
        for (i, f) in idx[:6].tolist():
            print("  oracle entity", i, "field", f, "%.9g" % st[i, f, 0], " == compact" if np.float32(st[i, f, 0]) == np.float32(float(sa[i, f, e])) else "", " == interp" if np.float32(st[i, f, 0]) == np.float32(float(sb[i, f, e])) else "")
        # how many contacts in that environment?  pairs within reach
        break
else:
    print("no divergence in 1500 held steps")
print(cp.compact_stats())
