"""A recording, in-memory stand-in for the `h5py` surface that gym-kmanip's episode logger uses.  TEST INFRASTRUCTURE.

`h5py` is absent from the build image and from the GPU box.  The reference's logger (gym_kmanip/log_h5py.py:13-61) and the
build's `EpisodeLogger` (h5py branch) make the same kind of calls -- `File(path, "w", rdcc_nbytes=...)`, `.attrs[k] = v`,
`create_group`, `create_dataset(name, shape | data=, dtype=, chunks=)`, `f[path][row] = value`, `flush`, `close` -- and this module
records them as a tree, so that the two trees can be compared node for node:
    tests/tools/refrun.py puts it into sys.modules as `h5py` before importing the reference's log_h5py (build container only);
    tests pass it to EpisodeLogger(h5py_module=...).
What is h5py's own behaviour (third-party, restated from its documentation, not from the reference):
    * create_dataset without dtype makes float32 ("dtype('f')"); with data= it takes the data's dtype;
    * an attribute value goes through numpy.asarray; object arrays (None, dataclass instances, dicts) have no HDF5 type ->
      TypeError; str / list of str are stored as variable-length strings; an empty list is an empty float64 attribute;
    * assigning into a row broadcasts like NumPy (a shape-(1,) value fills a whole row: log_h5py.py:55's `action` rows).
"""
from __future__ import annotations

import numpy as np

FILES = {}          # path -> File, in creation order (closed or not)


def _attr_record(value):
    arr = np.asarray(value)
    if arr.dtype.kind == "O":
        raise TypeError("Object dtype dtype('O') has no native HDF5 equivalent")
    if arr.dtype.kind in "US":
        return {"dtype": "str", "shape": list(arr.shape), "value": arr.tolist()}
    return {"dtype": str(arr.dtype), "shape": list(arr.shape), "value": arr.tolist()}


class Attrs(dict):
    def __setitem__(self, key, value):
        super().__setitem__(key, _attr_record(value))


class Dataset:
    def __init__(self, shape=None, dtype=None, data=None, chunks=None):
        if data is not None:
            data = np.asarray(data)
            self.array = data.astype(dtype) if dtype is not None else data.copy()
        else:
            self.array = np.zeros(tuple(shape), dtype=np.dtype("f") if dtype is None else np.dtype(dtype))
        self.chunks = None if chunks is None else tuple(int(c) for c in chunks)
        self.attrs = Attrs()

    shape = property(lambda self: self.array.shape)
    dtype = property(lambda self: self.array.dtype)

    def __setitem__(self, idx, value):
        self.array[idx] = value            # NumPy broadcasting + cast to the dataset's dtype, like h5py's write

    def __getitem__(self, idx):
        return self.array[idx]


class Group:
    def __init__(self):
        self.attrs = Attrs()
        self.children = {}

    def _walk(self, path, create):
        node = self
        for part in [p for p in path.strip("/").split("/") if p]:
            if part not in node.children:
                if not create:
                    raise KeyError(path)
                node.children[part] = Group()
            node = node.children[part]
        return node

    def create_group(self, path):
        parts = [p for p in path.strip("/").split("/") if p]
        node = self._walk("/".join(parts[:-1]), True)
        if parts[-1] in node.children:
            raise ValueError("Unable to create group (name already exists)")     # what h5py raises
        node.children[parts[-1]] = Group()
        return node.children[parts[-1]]

    def create_dataset(self, name, shape=None, dtype=None, data=None, chunks=None, **kw):
        parts = [p for p in name.strip("/").split("/") if p]
        node = self._walk("/".join(parts[:-1]), True)
        if parts[-1] in node.children:
            raise ValueError("Unable to create dataset (name already exists)")
        ds = Dataset(shape, dtype, data, chunks)
        node.children[parts[-1]] = ds
        return ds

    def __getitem__(self, path):
        return self._walk(path, False)

    def keys(self):
        return self.children.keys()


class File(Group):
    def __init__(self, path, mode="r", rdcc_nbytes=None, **kw):
        super().__init__()
        assert mode == "w", "the recorder only stands in for files opened for writing"
        self.path, self.mode, self.rdcc_nbytes = path, mode, rdcc_nbytes
        self.closed, self.flushes = False, 0
        FILES[path] = self

    def flush(self):
        self.flushes += 1

    def close(self):
        self.closed = True


def tree(node, skip_attr_values=()):
    """JSON-able description of a recorded file / group: attrs (dtype, shape, value), sub-groups, datasets (shape, dtype, chunks).
    Attribute VALUES listed in skip_attr_values (run-dependent ones: cpu_time, ...) are dropped, their dtype / shape kept."""
    out = {"attrs": {}, "groups": {}, "datasets": {}}
    for k, rec in node.attrs.items():
        rec = dict(rec)
        if k in skip_attr_values:
            rec.pop("value")
        out["attrs"][k] = rec
    for name, child in node.children.items():
        if isinstance(child, Dataset):
            out["datasets"][name] = {"shape": list(child.shape), "dtype": str(child.dtype),
                                     "chunks": None if child.chunks is None else list(child.chunks)}
        else:
            out["groups"][name] = tree(child, skip_attr_values)
    return out


def datasets(node, prefix=""):
    """{path: array} of every dataset below node."""
    out = {}
    for name, child in node.children.items():
        p = prefix + name
        if isinstance(child, Dataset):
            out[p] = child.array
        else:
            out.update(datasets(child, p + "/"))
    return out
