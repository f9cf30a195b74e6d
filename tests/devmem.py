"""Device tensors for the GPU tests, honouring the C ABI's stream contract (include/cpprob_hip.h): an engine works on its own
non-blocking stream, so a tensor torch fills on ITS stream must be complete before the call that hands it over.  Every helper
here returns only after torch's stream has finished producing the tensor."""
import numpy as np


def _done(t):
    import torch
    torch.cuda.current_stream(t.device).synchronize()
    return t


def dzeros(*shape, dtype=None, device="cuda"):
    import torch
    return _done(torch.zeros(*shape, dtype=dtype, device=device))


def dzeros_like(x):
    import torch
    return _done(torch.zeros_like(x))


def dtensor(a, dtype=None, device="cuda"):
    import torch
    return _done(torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device))


def dcat(parts, dtype=None, device="cuda"):
    import torch
    return _done(torch.cat(parts) if parts else torch.empty(0, dtype=dtype, device=device))
