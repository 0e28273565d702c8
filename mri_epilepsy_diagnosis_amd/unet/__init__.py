"""Drop-in for the third-party ``unet`` package used by the reference (``from unet import UNet``,
segmentation/routine.py:28)."""
from .unet import UNet, ConvolutionalBlock, EncodingBlock, DecodingBlock, Encoder, Decoder  # noqa: F401
