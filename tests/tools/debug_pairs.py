import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), '..', '..'), os.path.join(os.path.dirname(__file__), '..')]
import numpy as np, torch
from golden_util import load
from oracle.oracle import Oracle
from vectorizedmultiagentsimulator_amd.backend import HipWorld
name = sys.argv[1]; t = int(sys.argv[2]); lanes = int(sys.argv[3]); ent = int(sys.argv[4]); env = int(sys.argv[5]) if len(sys.argv) > 5 else 0
g = load(name); o = Oracle(g.spec)
st0 = np.ascontiguousarray(g.state0[t]).copy(); ft0 = np.ascontiguousarray(g.ft_in[t]).copy()
hw = HipWorld(g.spec, g.B, 'cuda:0', lanes_per_env=lanes)
words = (len(g.spec.pairs) + 31) // 32
np.set_printoptions(precision=6, suppress=True, linewidth=200)
for k, p in enumerate(g.spec.pairs):
    if ent not in (p.a, p.b): continue
    m = np.zeros(max(words, 1), np.uint32); m[k >> 5] = np.uint32(1 << (k & 31))
    want = st0.copy(); o.step(want, ft0.copy(), pair_mask=m)
    hw.state[:, :, :g.B].copy_(torch.from_numpy(st0)); hw.agent_ft[:g.spec.n_agents, :, :g.B].copy_(torch.from_numpy(ft0))
    hw.step(pair_mask=torch.from_numpy(m.view(np.int32)).cuda())
    got = hw.state[:, :, :g.B].cpu().numpy()
    print(k, (p.a, p.b, p.type), 'want', want[ent, 2:6, env], 'got', got[ent, 2:6, env], 'OK' if np.allclose(want[ent], got[ent], atol=1e-6) else 'DIFF')
