from ..unet import UNet  # noqa: F401
