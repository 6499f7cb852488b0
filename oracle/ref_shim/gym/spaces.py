"""Argument-storing stand-ins for gym.spaces.{Box,Discrete,MultiDiscrete,Tuple,Dict}."""
import numpy as np


class Space:
    shape = None
    dtype = None


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high = low, high
        if shape is None:
            shape = np.shape(low)
        self.shape, self.dtype = tuple(shape), dtype


class Discrete(Space):
    def __init__(self, n):
        self.n, self.shape, self.dtype = int(n), (), np.int64


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec)
        self.shape, self.dtype = self.nvec.shape, np.int64


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]
