from . import blob  # noqa: F401
from . import timer  # noqa: F401
