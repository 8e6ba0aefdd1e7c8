"""Optional torch-CPU kernels behind the oracle's heavy primitives.  TEST INFRASTRUCTURE (see oracle/__init__.py).

The oracle is a numpy restatement; numpy's elementwise passes (exp, erf, the GroupNorm reductions) run on one core, which
makes it a weak stand-in for "the reference's CPU path" when bench.py times it as ``cpu_baseline``.  The reference's own CPU
path is the same algorithm on PyTorch's ATen CPU kernels (oneDNN convolution, MKL sgemm, vectorised multi-threaded
elementwise ops), so the baseline leg may switch the five heavy primitives (conv2d, group_norm, silu, linear, gelu) to those
kernels.  Parity tests never do: they run the numpy forms, which are the ones pinned against the reference-generated fixtures
(tests/test_oracle_golden.py also checks that both backends agree)."""
_TORCH = False


def use_torch(flag=True):
    global _TORCH
    _TORCH = bool(flag)


def torch_on():
    return _TORCH
