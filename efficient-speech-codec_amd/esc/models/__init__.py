from .codecs import ESC, make_model, model_dict  # noqa: F401
from .discriminator import Discriminator  # noqa: F401
