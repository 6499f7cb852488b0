"""Test helpers around a LIVE reference world (oracle.ref): packing its state into the layout of include/vmas_hip.h,
and ``OracleBackend`` - a ``HipWorld`` look-alike that drives the CPU oracle on the adapter's packed buffers, so that
the drop-in plumbing (adapter.attach, core.World, the sharded rollout) can be exercised without a GPU.
TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch


def pack_state(world) -> np.ndarray:
    rows = []
    for e in world.entities:
        s = e.state
        rows.append(torch.cat([s.pos, s.vel, s.rot, s.ang_vel], dim=-1).T)  # [6, B]
    return torch.stack(rows).detach().cpu().numpy().astype(np.float32).copy()


def pack_ft(world) -> np.ndarray:
    rows = [torch.cat([a.state.force, a.state.torque], dim=-1).T for a in world.agents]  # [3, B]
    if not rows:
        return np.zeros((0, 3, world.batch_dim), np.float32)
    return torch.stack(rows).detach().cpu().numpy().astype(np.float32).copy()


def per_env_arrays(world, spec):
    """(joint_fixed_rot [J, B] | None, entity_gravity [E, 2, B] | None) of a reference world, spec order."""
    B = world.batch_dim
    ents = list(world.entities)
    jfr = None
    if any(j.per_env_fixed_rotation for j in spec.joints):
        jfr = np.zeros((len(spec.joints), B), np.float32)
        k = 0
        for ia, ea in enumerate(ents):
            for ib in range(ia + 1, len(ents)):
                j = world._joints.get(frozenset({ea.name, ents[ib].name}))
                if j is None:
                    continue
                fr = j.fixed_rotation
                jfr[k] = fr.reshape(B).numpy() if isinstance(fr, torch.Tensor) else float(fr)
                k += 1
    eg = None
    if any(e.per_env_gravity for e in spec.entities):
        eg = np.zeros((len(ents), 2, B), np.float32)
        for i, e in enumerate(ents):
            if spec.entities[i].per_env_gravity:
                eg[i] = e.gravity.T.numpy()
    return jfr, eg


class OracleBackend:
    """HipWorld look-alike driving the CPU oracle on caller-owned packed buffers (CPU tensors)."""

    def __init__(self, spec, batch, device, state, agent_ft):
        from oracle.oracle import Oracle

        self.o, self.spec, self.batch = Oracle(spec), spec, batch
        self.state, self.agent_ft = state, agent_ft
        self.ld = state.shape[-1]

    @staticmethod
    def _np(t):
        return None if t is None else t.numpy()

    def _ft(self):
        if self.spec.n_agents:
            return self.agent_ft.numpy()[: self.spec.n_agents]
        return np.zeros((0, 3, self.state.shape[-1]), np.float32)

    def step(self, pair_mask=None, joint_fixed_rot=None, entity_gravity=None, first_substep=0, n_substeps=0, stream=None):
        self.o.step(self.state.numpy(), self._ft(), batch=self.batch, joint_fixed_rot=self._np(joint_fixed_rot),
                    entity_gravity=self._np(entity_gravity))

    def step_exact(self, joint_fixed_rot=None, entity_gravity=None, stream=None):
        self.o.step_exact(self.state.numpy(), self._ft(), batch=self.batch, joint_fixed_rot=self._np(joint_fixed_rot),
                          entity_gravity=self._np(entity_gravity))

    def cast_rays(self, stream=None):
        return torch.from_numpy(self.o.cast_rays(self.state.numpy(), batch=self.batch))

    def pair_mask(self, stream=None):
        return torch.from_numpy(self.o.pair_mask(self.state.numpy(), self.batch).view(np.int32))

    def reserve_epilogue(self, *a):
        pass

    def exact_form(self):
        return 1  # (HipWorld.exact_form: the reference's rule inside the step call)

    def close(self):
        pass
