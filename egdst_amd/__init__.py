"""egdst_amd: MI355X-native DC-EGM solver + simulator behind the egdstmodel surface."""
from .model import egdstmodel, EgdstError  # noqa: F401
from .quadrature import quadpoints  # noqa: F401

__all__ = ['egdstmodel', 'EgdstError', 'quadpoints']
