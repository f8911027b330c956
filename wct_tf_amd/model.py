"""WCTModel: the test-mode graph of model.py:33-94 as a plain pipeline descriptor.

The reference builds one static TF graph; here the same attributes describe
what `wct_stylize` will run.  Placeholders become simple named slots that
`WCT.predict` fills (wct.py:97-103)."""
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
        if mode != 'test':
            # decoder training (model.py:178-223, train.py) is outside the stylize hot path
            raise NotImplementedError("only mode='test' is built on the MI355X path")
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
                decoded_encoded=None,
                pixel_loss=None, feature_loss=None, tv_loss=None, total_loss=None,
                train_op=None, learning_rate=None, global_step=None, summary_op=None))
        self.content_input = self.encoder_decoders[0].content_input
        self.decoded_output = self.encoder_decoders[-1].decoded
