"""WCTModel: the graph of model.py:33-223 as a plain pipeline descriptor.

The reference builds one static TF graph; here the same attributes describe
what the library runs.  mode='test': the stylize chain (`wct_stylize`); placeholders
become named slots that `WCT.predict` fills (wct.py:97-103).  mode='train': the
per-decoder training graph of model.py:178-223 -- its loss / train_op fields
name the parts of the ONE fused library call (`wct_train_step`), and
`WCTModel.train_step(sess, images, step)` runs it (what `sess.run(train_op, ...)`
is in train.py:156-160)."""
from collections import namedtuple

from .weights import RELU_CHANNELS, RELU_LEVEL, decoder_plan

# same field names as model.py:19-25; training-only fields are None in test mode
EncoderDecoder = namedtuple('EncoderDecoder',
                            'content_input content_encoder_model content_encoded '
                            'style_encoded '
                            'decoder_input, decoder_model decoded decoded_encoded '
                            'pixel_loss feature_loss tv_loss total_loss '
                            'train_op learning_rate global_step '
                            'summary_op')


class Slot(object):
    """Stand-in for a tf.placeholder_with_default."""

    def __init__(self, name, default=None):
        self.name = name
        self.default = default

    def __repr__(self):
        return 'Slot(%s)' % self.name


class WCTModel(object):
    def __init__(self, mode='test', relu_targets=['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1'],
                 vgg_path=None, *args, **kwargs):
        if mode not in ('test', 'train'):
            raise ValueError("mode must be 'train' or 'test' (model.py:33-40)")
        # training settings of model.py:123 (build_model's keyword arguments, passed through **kwargs by train.py:96-105)
        self.batch_size = kwargs.get('batch_size', 8)
        self.feature_weight = kwargs.get('feature_weight', 1)
        self.pixel_weight = kwargs.get('pixel_weight', 1)
        self.tv_weight = kwargs.get('tv_weight', 0)
        self.learning_rate = kwargs.get('learning_rate', 1e-4)
        self.lr_decay = kwargs.get('lr_decay', 5e-5)
        train = mode == 'train'
        for relu in relu_targets:
            if relu not in RELU_LEVEL:
                raise ValueError('unknown relu target %r' % (relu,))
        self.mode = mode
        self.relu_targets = list(relu_targets)
        self.vgg_path = vgg_path
        self.style_input = Slot('style_img')
        self.alpha = Slot('alpha', 1.)
        self.swap5 = Slot('swap5', False)
        self.ss_alpha = Slot('ss_alpha', .7)
        self.use_adain = Slot('use_adain', False)
        self.deepest_target = sorted(self.relu_targets)[-1]          # model.py:60
        self.vgg_model = {'target_layer': self.deepest_target, 'path': vgg_path}
        self.encoder_decoders = []
        for i, relu in enumerate(self.relu_targets):
            content_input = Slot('content_imgs') if i == 0 else 'clip(%s.decoded)' % self.relu_targets[i - 1]
            self.encoder_decoders.append(EncoderDecoder(
                content_input=content_input,
                content_encoder_model={'target_layer': relu, 'channels': RELU_CHANNELS[relu]},
                content_encoded=Slot('content_encoded_' + relu),
                style_encoded=Slot('style_encoded_' + relu),
                decoder_input=Slot('decoder_input_' + relu),
                decoder_model={'relu_target': relu, 'layers': decoder_plan(relu)},
                decoded=Slot('decoded_' + relu),
                # training-only fields (model.py:178-223): None for inference, as in the reference
                decoded_encoded=Slot('decoded_encoded_' + relu) if train else None,
                pixel_loss=Slot('pixel_loss_' + relu) if train else None,
                feature_loss=Slot('feature_loss_' + relu) if train else None,
                tv_loss=Slot('tv_loss_' + relu) if train else None,
                total_loss=Slot('total_loss_' + relu) if train else None,
                train_op={'call': 'wct_train_step', 'level': RELU_LEVEL[relu]} if train else None,
                learning_rate=Slot('learning_rate_' + relu, self.learning_rate) if train else None,
                global_step=Slot('global_step_train', 0) if train else None,
                summary_op=None))
        self.content_input = self.encoder_decoders[0].content_input
        self.decoded_output = self.encoder_decoders[-1].decoded

    def train_step(self, sess, images01, step, relu_target=None, opt_step=None):
        """One optimiser step of the decoder for `relu_target` (default: the first target) -- train.py:156-160's
        `sess.run([train_op, losses...])`.  sess: a wct_tf_amd.context.Context with encoder and decoder loaded;
        images01 [B][H][W][3] float32 in [0,1]; step: the global step BEFORE this update (learning-rate decay,
        ops.py:298-309); opt_step: steps since Adam's moments were zero (defaults to step + 1).
        Returns {'feature_loss', 'pixel_loss', 'tv_loss', 'total_loss'}."""
        if self.mode != 'train':
            raise RuntimeError("WCTModel(mode='test') has no train_op (model.py:206-208)")
        relu = relu_target or self.relu_targets[0]
        lr = self.learning_rate / (1.0 + step * self.lr_decay)
        return sess.train_step(relu, images01, step=(step + 1) if opt_step is None else opt_step, learning_rate=lr,
                               feature_weight=self.feature_weight, pixel_weight=self.pixel_weight, tv_weight=self.tv_weight)
