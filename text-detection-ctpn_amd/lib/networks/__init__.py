from .VGGnet_test import VGGnet_test  # noqa: F401
from . import factory  # noqa: F401
