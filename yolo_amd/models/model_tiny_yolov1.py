"""Drop-in for /root/reference/models/model_tiny_yolov1.py: same three public names, same call
signatures, built on the recording layers of yolo_amd.keras_facade and executed by libty1.so.

  ReshapeYOLO(target_shape)          :6-39   head activations (planar split, softmax / sigmoid, concat)
  original_backbone_network(inputs)  :42-90  8 frozen conv blocks + Flatten
  model_tiny_YOLOv1()                :93-99  -> (outputs, inputs)
"""
import torch

from ..keras_facade import (Layer, Conv2D, MaxPooling2D, Flatten, Dense, LeakyReLU, BatchNormalization,  # noqa: F401
                            Input, l2)
from .. import ops

# filters per conv block, as the reference hand-transcribed cfg/yolov1-tiny.cfg:24-110; the first
# six blocks are followed by a 2x2/2 max-pool
_FILTERS = (16, 32, 64, 128, 256, 512, 1024, 256)
_POOLED = 6


class ReshapeYOLO(Layer):
    """Keras-layer twin.  The reference hard-codes C=20, B=2 inside call(); here they are keyword
    arguments with those defaults (the C=1 copies in faces/ and airplanes/ only change C)."""

    def __init__(self, target_shape, C=20, B=2, **kwargs):
        super().__init__(**kwargs)
        self.target_shape = tuple(target_shape)
        self.C, self.B = C, B

    def compute_output_shape(self, input_shape):
        return (input_shape[0],) + self.target_shape

    def call(self, inputs, **kwargs):
        """inputs: (N, S*S*(C+5B)) planar Dense output on the GPU -> (N,S,S,C+5B)."""
        S = self.target_shape[0]
        z = inputs.to(torch.float32).contiguous()
        if z.shape[1] % 4:                                        # kernel wants a 16-byte row stride
            z = torch.nn.functional.pad(z, (0, 4 - z.shape[1] % 4))
        return ops.head_activate(z, None, S, self.B, self.C)

    def get_config(self):
        config = super().get_config()
        config.update({'target_shape': self.target_shape})
        return config


def original_backbone_network(inputs):
    x = inputs
    for i, filters in enumerate(_FILTERS):
        x = Conv2D(filters, (3, 3), padding='same', name=f'convolution_{i}', use_bias=False,
                   kernel_regularizer=l2(5e-4), trainable=False)(x)
        x = BatchNormalization(name=f'bn_convolution_{i}', trainable=False)(x)
        x = LeakyReLU(alpha=0.1)(x)
        if i < _POOLED:
            x = MaxPooling2D((2, 2), strides=(2, 2), padding='same')(x)
    return Flatten()(x)


def model_tiny_YOLOv1(C=20, B=2, S=7):
    inputs = Input((448, 448, 3))
    x = original_backbone_network(inputs)
    x = Dense(S * S * (C + 5 * B), activation='linear', name='connected_0')(x)
    outputs = ReshapeYOLO((S, S, C + 5 * B), C=C, B=B)(x)
    return outputs, inputs
