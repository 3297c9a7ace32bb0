"""vqvae_amd -- MI355X-native training hot path of dhgrs/chainer-VQ-VAE behind a
Chainer-shaped Chain / FunctionNode surface.  Host = Python; all arithmetic =
hand-written HIP (gfx950) in libvqvae_hip.so reached through a C ABI."""
from . import _lib, backend, core, functions, links, optimizers, reporting  # noqa: F401
from .core import (Chain, ChainList, FunctionNode, Link, Parameter, Variable,  # noqa: F401
                   config, report, using_config)
from .net import VAE, ConditionEmbed, Encoder  # noqa: F401
from .updaters import (VQVAE_ParallelUpdater, VQVAE_StandardUpdater,  # noqa: F401
                       concat_examples)
from .utils import VQ, ExponentialMovingAverage, MuLaw, StraightThrough, straight_through  # noqa: F401
from .wavenet import ResidualBlock, ResidualNet, WaveNet  # noqa: F401
from .synthesis import synthesize  # noqa: F401
from .reporting import LogReport, PlotReport, PrintReport  # noqa: F401
