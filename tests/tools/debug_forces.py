import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), '..', '..'), os.path.join(os.path.dirname(__file__), '..')]
import numpy as np, torch
from golden_util import load
from oracle.oracle import Oracle
from vectorizedmultiagentsimulator_amd.backend import HipWorld
name = sys.argv[1] if len(sys.argv) > 1 else 'balance_n3'
t = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = load(name); o = Oracle(g.spec)
st0 = np.ascontiguousarray(g.state0[t]).copy(); ft0 = np.ascontiguousarray(g.ft_in[t]).copy()
want = st0.copy(); o.step(want, ft0.copy())
hw = HipWorld(g.spec, g.B, 'cuda:0', lanes_per_env=lanes)
hw.state[:, :, :g.B].copy_(torch.from_numpy(st0)); hw.agent_ft[:g.spec.n_agents, :, :g.B].copy_(torch.from_numpy(ft0))
hw.step()
got = hw.state[:, :, :g.B].cpu().numpy()
env = 0
for i, e in enumerate(g.spec.entities):
    if not (e.flags & 3): continue
    dv_w = (want[i, 2:4, env] - st0[i, 2:4, env] * e.one_minus_drag) / g.spec.sub_dt * e.mass
    dv_g = (got[i, 2:4, env] - st0[i, 2:4, env] * e.one_minus_drag) / g.spec.sub_dt * e.mass
    print(i, e.name, 'F oracle', dv_w, 'F hip', dv_g)
np.set_printoptions(precision=6, suppress=True, linewidth=200)
for i, e in enumerate(g.spec.entities):
    if not (e.flags & 3): continue
    print(i, e.name, 'st0 ', st0[i, :, env]); print('   want', want[i, :, env]); print('   got ', got[i, :, env])
