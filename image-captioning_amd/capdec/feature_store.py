"""Precomputed-feature store: one memory-mapped fp32 matrix instead of one pickle per image.

The reference writes ``precomputed/{arch}/{dataset}/{image}.pkl`` (train.py:37-93) and reads them back one by one in
its Dataset (get_loader.py:163-169,186-189); at tens of thousands of samples per second on the GPU that loader is the
bottleneck.  Here the features of a split live in ``features.f32`` ([N,D] row-major, memory-mapped) next to
``index.json`` (image id -> row), and batches reach the GPU through two pinned staging buffers and a copy stream, so
the host gather of batch k+1 and its H2D copy overlap the GPU work on batch k.
"""
import json
import os

import numpy as np
import torch


class FeatureStore:
    def __init__(self, root, mode="r"):
        self.root = root
        with open(os.path.join(root, "index.json")) as fh:
            meta = json.load(fh)
        self.ids = meta["ids"]
        self.dim = int(meta["dim"])
        self.row_of = {k: i for i, k in enumerate(self.ids)}
        self.data = np.memmap(os.path.join(root, "features.f32"), dtype=np.float32, mode=mode,
                              shape=(len(self.ids), self.dim))

    def __len__(self):
        return len(self.ids)

    # ---- building ---------------------------------------------------------------------------------------
    @staticmethod
    def create(root, ids, dim):
        """Allocate an empty store for ``ids`` (list of image ids) x ``dim``; fill ``store.data[row]`` and flush."""
        os.makedirs(root, exist_ok=True)
        ids = [str(i) for i in ids]
        if len(set(ids)) != len(ids):
            raise ValueError("feature store: duplicate image ids")
        with open(os.path.join(root, "index.json"), "w") as fh:
            json.dump({"ids": ids, "dim": int(dim), "dtype": "float32"}, fh)
        mm = np.memmap(os.path.join(root, "features.f32"), dtype=np.float32, mode="w+", shape=(len(ids), int(dim)))
        del mm
        return FeatureStore(root, mode="r+")

    @staticmethod
    def from_arrays(root, ids, features):
        """features: [N,D] array-like (numpy or CPU tensor)."""
        feats = np.asarray(features, dtype=np.float32)
        store = FeatureStore.create(root, ids, feats.shape[1])
        store.data[:] = feats
        store.data.flush()
        return store

    # ---- reading ----------------------------------------------------------------------------------------
    def rows(self, ids):
        """Row indices of a list of image ids."""
        return np.fromiter((self.row_of[str(i)] for i in ids), dtype=np.int64, count=len(ids))

    def gather(self, rows, out=None):
        """Host gather of ``rows`` into a (pinned, when CUDA is available) [len(rows), D] tensor."""
        if out is None:
            out = torch.empty((len(rows), self.dim), dtype=torch.float32, pin_memory=torch.cuda.is_available())
        np.take(self.data, rows, axis=0, out=out.numpy()[: len(rows)])
        return out[: len(rows)]

    def batches(self, batch_size, device, order=None, drop_last=False):
        """Yield (rows, features on ``device``) over ``order`` (default: all rows in storage order).  Double-buffered:
        while the consumer works on one batch, the next is gathered into the other pinned buffer and copied on a
        side stream; the consumer's stream waits on that copy's event before the batch is handed out."""
        order = np.arange(len(self), dtype=np.int64) if order is None else np.asarray(order, dtype=np.int64)
        starts = list(range(0, len(order), batch_size))
        if drop_last and starts and len(order) - starts[-1] < batch_size:
            starts.pop()
        dev = torch.device(device)
        if dev.type != "cuda":
            for s in starts:
                r = order[s:s + batch_size]
                yield r, self.gather(r).clone()
            return
        copy_stream = torch.cuda.Stream(device=dev)
        host = [torch.empty((batch_size, self.dim), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        devb = [torch.empty((batch_size, self.dim), dtype=torch.float32, device=dev) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        free = [torch.cuda.Event() for _ in range(2)]

        def stage(k, s):
            r = order[s:s + batch_size]
            slot = k & 1
            free[slot].synchronize()                       # the consumer is done with this slot's previous batch
            self.gather(r, out=host[slot])
            with torch.cuda.stream(copy_stream):
                devb[slot][: len(r)].copy_(host[slot][: len(r)], non_blocking=True)
                ready[slot].record(copy_stream)
            return r

        for slot in range(2):
            free[slot].record()
        pending = stage(0, starts[0]) if starts else None
        for k, s in enumerate(starts):
            r = pending
            if k + 1 < len(starts):
                pending = stage(k + 1, starts[k + 1])
            slot = k & 1
            torch.cuda.current_stream(dev).wait_event(ready[slot])
            yield r, devb[slot][: len(r)]
            free[slot].record()                            # recorded on the consumer's stream after its use


class _TensorOnlyUnpickler:
    """Reads the reference's per-image ``pickle.dump(tensor)`` files without running anything but the handful of
    torch tensor-rebuild callables they contain."""

    @staticmethod
    def load(path):
        import io
        import pickle

        allowed = {("torch._utils", "_rebuild_tensor_v2"), ("torch.storage", "_load_from_bytes"),
                   ("collections", "OrderedDict"), ("torch", "FloatStorage"), ("torch", "Size")}

        def safe_load_from_bytes(b):
            return torch.load(io.BytesIO(b), weights_only=True)

        class U(pickle.Unpickler):
            def find_class(self, module, name):
                if (module, name) == ("torch.storage", "_load_from_bytes"):
                    return safe_load_from_bytes
                if (module, name) in allowed:
                    return getattr(__import__(module, fromlist=[name]), name)
                raise pickle.UnpicklingError(f"feature pickle refers to {module}.{name}: refused")

        with open(path, "rb") as fh:
            return U(fh).load()


def _feature_vector(path):
    """One per-image feature file as a flat fp32 numpy vector; anything but a floating-point tensor is refused."""
    obj = _TensorOnlyUnpickler.load(path)
    if not isinstance(obj, torch.Tensor) or not obj.is_floating_point():
        raise ValueError(f"{path}: expected a floating-point tensor, found {type(obj).__name__}"
                         + (f" of dtype {obj.dtype}" if isinstance(obj, torch.Tensor) else ""))
    return obj.detach().to(torch.float32).reshape(-1).contiguous().numpy()


def import_pickle_dir(src_dir, dst_root):
    """Convert a directory of the reference's ``{image}.pkl`` feature files into a store.

    The directory must be the output of YOUR OWN precompute run (reference train.py:37-93): the files are read with a
    pickle machine restricted to the tensor-rebuild callables such files contain (anything else aborts the import), which
    is a guard against accidents, not a sandbox for files of unknown origin.  Only regular files directly under
    ``src_dir`` are read (no symbolic links, no sub-directories); every file must hold one floating-point tensor, stored
    as fp32, and all of them the same number of elements."""
    src_dir = os.path.abspath(src_dir)
    names = sorted(n for n in os.listdir(src_dir)
                   if n.endswith(".pkl") and os.path.isfile(os.path.join(src_dir, n)) and not os.path.islink(os.path.join(src_dir, n)))
    if not names:
        raise ValueError(f"no .pkl files under {src_dir}")
    first = _feature_vector(os.path.join(src_dir, names[0]))
    store = FeatureStore.create(dst_root, [n[:-4] for n in names], first.size)
    for i, n in enumerate(names):
        vec = first if i == 0 else _feature_vector(os.path.join(src_dir, n))
        if vec.size != first.size:
            raise ValueError(f"{n}: {vec.size} elements, the first file has {first.size}")
        store.data[i] = vec
    store.data.flush()
    return store
