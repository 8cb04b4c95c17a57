from .generators import ImplicitGenerator3d  # noqa: F401
from . import siren, volumetric_rendering, math_utils_torch  # noqa: F401
