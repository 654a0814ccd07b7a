from . import conv  # noqa
