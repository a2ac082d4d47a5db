"""The driver's round-end smoke run as a test: __graft_entry__.smoke() -- one training step of deeplabv3plus_resnet50 on
cuda:0 against the CPU oracle -- must not be able to rot while the rest of the suite stays green (it did once: a stem kernel
with a different rounding moved max-pool near-ties, and nothing here ran smoke())."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU path")
    import __graft_entry__
    __graft_entry__.smoke()
