"""The layers reachable from a backbone `keras_layer { class_name: ... }` block
(reference easy_rec/python/layers/keras/__init__.py; located by utils/load_class.load_keras_layer)."""
from .blocks import MLP, Add  # noqa: F401
from .interaction import CIN, FM, Cross, DotInteraction  # noqa: F401
from .multi_task import MMoE  # noqa: F401
from .din import DIN  # noqa: F401
from .fibinet import SENet  # noqa: F401
from .standard import Activation, Dense, Dropout  # noqa: F401
