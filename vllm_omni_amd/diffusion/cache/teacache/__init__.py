from .backend import TeaCacheBackend
from .config import TeaCacheConfig
from .native import TeaCacheDeviceState

__all__ = ["TeaCacheBackend", "TeaCacheConfig", "TeaCacheDeviceState"]
