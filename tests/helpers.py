"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from oracle import mimo_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def state_from(fx, prefix):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith(prefix)}


def cfg_from_meta(meta):
    Ci, Co, S, f = (int(v) for v in meta[:4])
    return O.NetConfig(in_channels=Ci, out_channels=Co, num_subnetworks=S, filter_base_count=f)


def rel_err(a, b):
    """max |a-b| / max(|b|) — error relative to the tensor's scale (the fp32
    tolerance the north_star states is 1e-3 on this measure)."""
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    denom = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / denom


def rms_rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (rendezvous of the multi-process tests)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def report(*parts):
    """Print an observed-error line and, when MIMO_PARITY_LOG names a file, append it there (the GPU run's
    parity_errors.txt that is committed under profiles/)."""
    msg = " ".join(str(p) for p in parts)
    print(msg)
    path = os.environ.get("MIMO_PARITY_LOG")
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
        with open(path, "a") as fh:
            fh.write(f"{test}: {msg}\n")
