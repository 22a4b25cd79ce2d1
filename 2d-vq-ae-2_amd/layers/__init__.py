from . import conv, conv_block, vq  # noqa: F401
