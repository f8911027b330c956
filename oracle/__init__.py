"""CPU oracle for the WCT hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in NumPy, the arithmetic of the reference's stylize path
(eridgd/WCT-TF).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker -- never as
the thing measured or shipped.  The product (``wct_tf_amd``) must not import it.

Parity status:
  * ``wct_np`` and ``coral_numpy`` restatements are PINNED: checked against
    outputs of the reference's own functions executed in the build container
    (``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
  * ``wct_tf``, ``adain`` and the conv / pool / upsample / encoder / decoder
    restatements are "parity unpinned" by the reference (it ships no tests and
    TensorFlow/Keras are not installable here); they are cross-checked against an
    independent torch-CPU implementation in ``tests/test_oracle.py``.
"""
from .wct_oracle import (wct_np, wct_tf, adain, coral_numpy, mat_sqrt_numpy,
                         preserve_colors_np, style_swap, wct_style_swap)
from .net_oracle import (conv3x3_reflect, conv3x3_reflect_wino_f16, maxpool2x2_same, upsample2x_nearest,
                         encode, decode, stylize, preprocess, postprocess,
                         ENCODER_LAYERS, DECODER_ARCHS, decoder_layers)
