"""TEST INFRASTRUCTURE -- runs the reference's torch code with CORRECTLY ROUNDED elementary functions.

torch's sin / cos / atan2 / asin / exp (SLEEF's 1-ulp kernels for full vectors, libm for the tails) and, in this
AVX-512 CPU build, even sqrt (0.6 % of results are 1 ulp off) are not correctly rounded, and the CUDA build's differ
again: the reference's own numbers carry an implementation-dependent last bit.  The oracle and the HIP kernels evaluate
those functions correctly rounded (float64 inside, rounded once; oracle/oracle_math.h).  To pin EVERYTHING ELSE of the
restatement bit for bit -- operation order, fused multiply-adds inside torch.cross / torch.norm, scalar / tensor
= reciprocal * scalar, python-double scalar arithmetic -- the golden generators can be run in "CR mode"
(`python oracle/gen_golden.py --cr` -> tests/golden/cr/): the reference's code, unmodified, with these six torch
functions routed through float64 (sin, cos, asin, atan2, exp of a float32 via float64 and rounded once are correctly
rounded up to double rounding, ~1e-9 of the arguments; sqrt via float64 is always correctly rounded).

TorchScript compiles `@torch.jit.script` functions to aten calls that a Python-level patch cannot reach, so the generators
set PYTORCH_JIT=0 BEFORE torch is imported (scripted functions then run as plain Python).  tests/test_oracle_vs_reference.py
checks that the ordinary goldens come out bit-identical with TorchScript ON, i.e. that this does not change what the
reference computes.

The patch is switched at run time (`with cr_torch.correctly_rounded(): ...`), so that the ordinary generator can ALSO record,
next to each of the reference's outputs, what the same code returns on the same recorded inputs with correctly rounded
functions (`*_cr` arrays of tests/golden/step_*.npz): the reference's own last-bit freedom on that very sample.  Where it
exceeds the parity gate (the motor model's sqrt next to zero thrust amplifies one ulp of a sine to 2e-5 rad/s of body
rate), no implementation can be within the gate of "the" reference, and the tests then ask for the correctly rounded answer.
"""
import os
import sys

import contextlib

UNARY = ("sin", "cos", "asin", "arcsin", "exp", "sqrt")
_ENABLED = False


def enable(on=True):
    global _ENABLED
    _ENABLED = bool(on)


@contextlib.contextmanager
def correctly_rounded(on=True):
    global _ENABLED
    prev, _ENABLED = _ENABLED, bool(on)
    try:
        yield
    finally:
        _ENABLED = prev


def requested():
    return "--cr" in sys.argv or os.environ.get("AGX_GOLDEN_CR") == "1"


def prepare_environment():
    """call BEFORE `import torch`"""
    if "torch" in sys.modules:
        raise RuntimeError("cr_torch.prepare_environment() must run before torch is imported (PYTORCH_JIT=0)")
    os.environ["PYTORCH_JIT"] = "0"


def install():
    import torch

    if getattr(torch, "_agx_cr_installed", False):
        return
    if os.environ.get("PYTORCH_JIT") != "0":
        raise RuntimeError("CR mode needs PYTORCH_JIT=0 (cr_torch.prepare_environment() before importing torch)")

    def unary(name):
        orig = getattr(torch, name)

        def f(x, *a, **k):
            if _ENABLED and isinstance(x, torch.Tensor) and x.dtype == torch.float32 and not a and not k:
                return orig(x.double()).float()
            return orig(x, *a, **k)

        f.__name__ = name
        return f

    for name in UNARY:
        f = unary(name)
        setattr(torch, name, f)
        setattr(torch.Tensor, name, f)
    orig_atan2 = torch.atan2

    def atan2(y, x):
        if _ENABLED and y.dtype == torch.float32 and x.dtype == torch.float32:
            return orig_atan2(y.double(), x.double()).float()
        return orig_atan2(y, x)

    torch.atan2 = torch.arctan2 = atan2
    torch.Tensor.atan2 = torch.Tensor.arctan2 = atan2
    torch._agx_cr_installed = True
