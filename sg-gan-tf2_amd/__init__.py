"""sggan_amd -- MI355X-native SG-GAN train-step hot path.

The directory is called ``sg-gan-tf2_amd`` (not an importable name); import it as
``sggan_amd`` via the shim at the repository root.
"""
from . import _abi  # noqa: F401
from ._abi import SggError, lib  # noqa: F401
from .model import default_args, sggan  # noqa: F401
from .module import Discriminator, Generator, discriminator, generator_resnet  # noqa: F401
from .ops import conv2d, deconv2d, instance_norm, lrelu, relu, tanh  # noqa: F401

__all__ = ["sggan", "default_args", "generator_resnet", "discriminator", "Generator", "Discriminator",
           "conv2d", "deconv2d", "instance_norm", "lrelu", "relu", "tanh", "lib", "SggError", "LIB_PATH"]


def __getattr__(name):
    if name == "LIB_PATH":              # the library actually bound (tools/ may select another build before first use)
        return _abi.LIB_PATH
    raise AttributeError(name)
