"""TEST INFRASTRUCTURE — an in-memory stand-in for the three h5py calls the predictors make (h5py is not installed here nor on
the GPU box): h5py.File(path, mode) as a context manager, File.create_dataset(name, data=... | shape=, dtype=...), File[name].
Files live in the module-level STORE dict keyed by path."""
import sys
import types

import numpy as np

STORE = {}


class File:
    def __init__(self, path, mode="r"):
        self.path = str(path)
        if "w" in mode:
            STORE[self.path] = {}
        elif self.path not in STORE:
            raise OSError(f"no such in-memory file: {self.path}")
        self.d = STORE[self.path]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def create_dataset(self, name, data=None, shape=None, dtype=None, **_kw):
        self.d[name] = np.array(data) if data is not None else np.zeros(shape, dtype=dtype)
        return self.d[name]

    def __getitem__(self, name):
        return self.d[name]

    def __contains__(self, name):
        return name in self.d

    def keys(self):
        return self.d.keys()


def install():
    mod = types.ModuleType("h5py")
    mod.File = File
    mod.Dataset = np.ndarray
    sys.modules["h5py"] = mod
    return mod
