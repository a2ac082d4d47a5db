"""Small shims for torch classes the reference instantiates but that carry no state."""
import torch.nn as nn

try:
    from torch.ao.nn.quantized import FloatFunctional  # noqa: F401  (reference: nn.quantized.FloatFunctional)
except Exception:  # pragma: no cover
    class FloatFunctional(nn.Module):
        def add(self, a, b):
            return a + b

try:
    from torch.ao.quantization import DeQuantStub, QuantStub  # noqa: F401
except Exception:  # pragma: no cover
    class QuantStub(nn.Identity):
        pass

    class DeQuantStub(nn.Identity):
        pass
