"""nexoclom_amd: MI355X-native implementation of nexoclom's particle-tracking + image hot path.

Same front door as the reference package (nexoclom/__init__.py:9-14): Input, Output, ModelImage,
SSObject.  Importing this package touches neither a database nor the GPU; the HIP library is
loaded on first use and there is no CPU fallback for the integrator or the image kernels.
"""
__version__ = '0.1.0'

from .Input import Input                    # noqa: F401
from .Output import Output                  # noqa: F401
from .ModelImage import ModelImage, ModelResult   # noqa: F401
from .LOSResult import LOSResult, SpacecraftData   # noqa: F401
from .solarsystem import SSObject, planet_dist    # noqa: F401
from .atomicdata import gValue, RadPresConst, PhotoRate, atomicmass   # noqa: F401
from .input_classes import InputError       # noqa: F401
