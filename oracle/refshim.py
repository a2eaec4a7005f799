"""TEST INFRASTRUCTURE ONLY — loader for the *live* upstream reference.

Imports the reference's own modules from ``/root/reference/{imdb-wiki,agedb}-dir``
so that (a) ``tests/golden/gen_golden.py`` can produce golden vectors and
(b) CPU tests can pin ``oracle/`` against the reference when it is present.
``/root/reference`` does not exist on the GPU box: everything here degrades to
``available() == False`` there and nothing under ``-m gpu``, ``smoke()`` or
``bench.py`` may call into this file.

The three shims are the ones SURVEY.md §8(c) verified:
  1. ``torch.Tensor.cuda`` -> identity while reference code runs
     (reference ``fds.py:52`` ends in ``.cuda()``),
  2. stub ``torchvision`` / ``torchvision.transforms`` (reference ``datasets.py:7``),
  3. stub ``tensorboard_logger`` (reference ``train.py:12``; unused here).
The reference's flat module names (``utils``, ``loss``, ``fds`` ...) are loaded
under private names and removed from ``sys.modules`` again, so they never shadow
the drop-in modules of the same names in this repo.
"""
import contextlib
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DIR_REFERENCE_ROOT", "/root/reference")


def available(subdir="imdb-wiki-dir"):
    return os.path.isfile(os.path.join(REFERENCE_ROOT, subdir, "fds.py"))


@contextlib.contextmanager
def cuda_identity():
    """Make ``Tensor.cuda()`` / ``Module.cuda()`` no-ops while reference code runs on CPU."""
    import torch
    t_old = torch.Tensor.cuda
    m_old = torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = t_old
        torch.nn.Module.cuda = m_old


_CACHE = {}


def load(subdir="imdb-wiki-dir"):
    """Return a namespace with the reference modules fds, loss, utils, resnet, datasets."""
    if subdir in _CACHE:
        return _CACHE[subdir]
    if not available(subdir):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}/{subdir}")
    base = os.path.join(REFERENCE_ROOT, subdir)
    names = ["utils", "loss", "fds", "resnet", "datasets"]
    saved = {n: sys.modules.get(n) for n in names + ["torchvision", "torchvision.transforms"]}
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    ns = types.SimpleNamespace()
    try:
        for n in names:
            spec = importlib.util.spec_from_file_location(n, os.path.join(base, n + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[n] = mod          # so `from utils import ...` inside fds.py resolves
            spec.loader.exec_module(mod)
            setattr(ns, n, mod)
    finally:
        for n, old in saved.items():
            if old is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = old
    _CACHE[subdir] = ns
    return ns


def make_fds(subdir="imdb-wiki-dir", **kw):
    """Construct the reference FDS on CPU."""
    ref = load(subdir)
    with cuda_identity():
        return ref.fds.FDS(**kw)


def make_resnet50(subdir="imdb-wiki-dir", **kw):
    ref = load(subdir)
    with cuda_identity():
        return ref.resnet.resnet50(**kw)


def prepare_weights(labels, subdir="imdb-wiki-dir", **kw):
    """Run the reference ``_prepare_weights`` (datasets.py:55-83) on a bare label list."""
    import pandas as pd
    ref = load(subdir)
    cls = ref.datasets.IMDBWIKI if hasattr(ref.datasets, "IMDBWIKI") else ref.datasets.AgeDB
    obj = cls.__new__(cls)
    obj.df = pd.DataFrame({"age": labels})
    return obj._prepare_weights(**kw)


def load_stsb():
    """Reference sts-b-dir FDS pieces only (fds.py + util.py; both import nothing beyond numpy/scipy/torch)."""
    if "sts-b" in _CACHE:
        return _CACHE["sts-b"]
    base = os.path.join(REFERENCE_ROOT, "sts-b-dir")
    if not os.path.isfile(os.path.join(base, "fds.py")):
        raise RuntimeError("reference sts-b-dir not present")
    saved = {n: sys.modules.get(n) for n in ("util", "fds")}
    ns = types.SimpleNamespace()
    try:
        for n in ("util", "fds"):
            spec = importlib.util.spec_from_file_location(n, os.path.join(base, n + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[n] = mod
            spec.loader.exec_module(mod)
            setattr(ns, n, mod)
    finally:
        for n, old in saved.items():
            if old is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = old
    _CACHE["sts-b"] = ns
    return ns


@contextlib.contextmanager
def device_transfer_clones():
    """NYUD2 shim: ``Tensor.cuda()`` / ``Tensor.cpu()`` return a NEW tensor (clone), like a real device transfer does
    (nyud2-dir/models/fds.py:88-96 relies on it: the re-created running buffers stop aliasing the *_last_epoch ones), and
    ``np.bool`` (removed in numpy >= 1.24, used at fds.py:110) is restored for the duration."""
    import numpy as np
    import torch
    t_cuda, t_cpu, m_cuda = torch.Tensor.cuda, torch.Tensor.cpu, torch.nn.Module.cuda
    had_bool = hasattr(np, "bool")
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if not had_bool:
        np.bool = bool
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.Tensor.cpu, torch.nn.Module.cuda = t_cuda, t_cpu, m_cuda
        if not had_bool:
            del np.bool


def load_nyud2():
    """Reference nyud2-dir FDS pieces only (models/fds.py + util.py)."""
    if "nyud2" in _CACHE:
        return _CACHE["nyud2"]
    base = os.path.join(REFERENCE_ROOT, "nyud2-dir")
    if not os.path.isfile(os.path.join(base, "models", "fds.py")):
        raise RuntimeError("reference nyud2-dir not present")
    saved = {n: sys.modules.get(n) for n in ("util", "fds")}
    ns = types.SimpleNamespace()
    try:
        for n, rel in (("util", "util.py"), ("fds", os.path.join("models", "fds.py"))):
            spec = importlib.util.spec_from_file_location(n, os.path.join(base, rel))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[n] = mod
            spec.loader.exec_module(mod)
            setattr(ns, n, mod)
    finally:
        for n, old in saved.items():
            if old is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = old
    _CACHE["nyud2"] = ns
    return ns
