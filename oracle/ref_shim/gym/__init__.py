"""Minimal stand-in for the `gym` package (absent in this image, no network).

TEST INFRASTRUCTURE ONLY.  The reference (`/root/reference/vmas`) imports
`gym.spaces` at module import time (vmas/simulator/environment/environment.py:14,
vmas/simulator/environment/gym/gym.py:7).  Only constructors that store their
arguments are needed to build environments and step them; nothing here is on the
product path.
"""
from . import spaces  # noqa: F401


class Env:  # gym.Env placeholder (wrappers are never instantiated by our tools)
    metadata = {}


class vector:  # noqa: N801 - namespace placeholder for `gym.vector.VectorEnv`
    class VectorEnv:
        pass
