from . import resnet  # noqa: F401  (the reference also imports xception, which no entry point can select)
