"""Load the reference's ``models.py`` on CPU in the BUILD CONTAINER ONLY.

TEST-FIXTURE TOOLING.  Used by ``oracle/make_golden.py`` alone; nothing that runs
on the GPU box imports this (``/root/reference`` does not exist there).

The reference imports tensorflow, pytorch_lightning and torchmetrics (absent
here) and calls ``.cuda()`` in constructors (models.py:199,237,739-742).  We
pre-seed ``sys.modules`` with inert stand-ins and make ``.cuda()`` the identity,
then import the reference's own file unmodified.  Bytecode writing is disabled so
the read-only reference tree is not touched.
"""
from __future__ import annotations

import ast
import sys
import types

REFERENCE_ROOT = "/root/reference"


def install():
    sys.dont_write_bytecode = True
    import numpy as np
    import torch
    from torch import nn

    class _TfTensor:
        def __init__(self, a):
            self.a = np.asarray(a, dtype=np.float32)

        def numpy(self):
            return self.a

    tf = types.ModuleType("tensorflow")
    tf.convert_to_tensor = lambda rows: _TfTensor([list(r) for r in rows])
    tf.cast = lambda x, dt: list(x)
    tf.float32, tf.int32 = "float32", "int32"
    sys.modules["tensorflow"] = tf

    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    sys.modules["pytorch_lightning"] = pl
    for name, attrs in [("pytorch_lightning.loggers", ["TensorBoardLogger"]),
                        ("pytorch_lightning.callbacks", ["ModelCheckpoint"]),
                        ("pytorch_lightning.callbacks.early_stopping", ["EarlyStopping"])]:
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, type(a, (), {}))
        sys.modules[name] = m

    class Accuracy:            # torchmetrics.Accuracy()(pred_idx, target_idx) -> mean(pred == target)
        def cuda(self):
            return self

        def __call__(self, pred, target):
            return (pred == target).float().mean()

    tm = types.ModuleType("torchmetrics")
    tm.Accuracy = Accuracy
    sys.modules["torchmetrics"] = tm

    # (torch.utils.checkpoint -- the reference's dense layers use it -- imports torch._dynamo, which calls importlib's find_spec on a list of
    # module names: a stub without a __spec__ makes that raise)
    import importlib.machinery
    for name in ("tensorflow", "pytorch_lightning", "pytorch_lightning.loggers", "pytorch_lightning.callbacks",
                 "pytorch_lightning.callbacks.early_stopping", "torchmetrics"):
        sys.modules[name].__spec__ = importlib.machinery.ModuleSpec(name, None)

    nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import models  # noqa: E402  (the reference's models.py)
    return models


def reference_functions(path, names):
    """exec only the named top-level ``def``s of a reference script.

    ``equivariance_test.py`` cannot be imported (it loads a wav at import time,
    :109); its two shift helpers (:122-146) are pure functions of torch tensors.
    """
    import torch
    src = open(path).read()
    tree = ast.parse(src)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=picked, type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return {n: ns[n] for n in names}
