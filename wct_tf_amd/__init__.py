"""MI355X-native stylize path with the eridgd/WCT-TF surface (WCT, WCTModel, wct_np, ...).

Importing the package never loads the HIP library; the first call does, and
fails loudly if libwct_hip.so is missing (no CPU fallback)."""
from .weights import (synthetic_weights, synthetic_image, synthetic_features, save_weights,
                      load_weights, RELU_TARGETS)
from .model import WCTModel, EncoderDecoder
from .wct import WCT
from .ops import wct_np, wct_tf, adain, coral_numpy, preserve_colors_np
from .context import Context, default_context

__all__ = ['WCT', 'WCTModel', 'EncoderDecoder', 'wct_np', 'wct_tf', 'adain', 'coral_numpy',
           'preserve_colors_np', 'Context', 'default_context', 'synthetic_weights',
           'synthetic_image', 'synthetic_features', 'save_weights', 'load_weights', 'RELU_TARGETS']
