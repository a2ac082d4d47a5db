"""Drop-in counterpart of the reference's ``network`` package (construction API of
network/modeling.py, network/_deeplab.py, network/utils.py, network/backbone/resnet.py)
running on hand-written gfx950 kernels.  The reference's own ``network/__init__.py``
is empty, which makes ``network.modeling`` at train.py:284 an AttributeError; importing
``modeling`` here is the one deliberate difference."""
from . import modeling  # noqa: F401
from .modeling import *  # noqa: F401,F403
