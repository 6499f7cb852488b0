"""Lidar sensor (vmas/simulator/sensors.py:47-123) on top of the fused ray-cast kernel.

All Lidars of all agents are cast by ONE kernel launch (``World.cast_rays_all``); a
``Lidar.measure()`` slices its rows out of the result of the launch made for the current
step (cached per world state version by the caller, see Environment).
"""
from __future__ import annotations

from typing import Callable

import torch


class Lidar:
    def __init__(self, world, angle_start: float = 0.0, angle_end: float = 2 * torch.pi, n_rays: int = 8,
                 max_range: float = 1.0, entity_filter: Callable = lambda _: True, render_color=None,
                 alpha: float = 1.0, render: bool = True):
        self._world = world
        self.agent = None
        if (angle_start - angle_end) % (torch.pi * 2) < 1e-5:  # sensors.py:61-70
            angles = torch.linspace(angle_start, angle_end, n_rays + 1)[:n_rays]
        else:
            angles = torch.linspace(angle_start, angle_end, n_rays)
        self._angles = angles  # [n_rays]; identical for every environment
        self._max_range = max_range
        self._entity_filter = entity_filter
        self._last_measurement = None

    @property
    def entity_filter(self):
        return self._entity_filter

    @entity_filter.setter
    def entity_filter(self, f):
        self._entity_filter = f
        self._world._invalidate_backend()

    def _index(self) -> int:
        k = 0
        for a in self._world.agents:
            for s in a.sensors:
                if s is self:
                    return k
                if hasattr(s, "_angles"):
                    k += 1
        raise AssertionError("sensor is not attached to an agent of its world")

    def measure(self, all_rays: torch.Tensor = None) -> torch.Tensor:
        """[B, n_rays] distances (sensors.py:101-123).  ``all_rays`` may carry the result of
        a ``World.cast_rays_all()`` already made for this state."""
        if all_rays is None:
            all_rays = self._world.cast_rays_all()
        m = all_rays[self._index(), : self._angles.shape[0], : self._world.batch_dim].T
        self._last_measurement = m
        return m
