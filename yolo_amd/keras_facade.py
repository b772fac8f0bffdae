"""The slice of the Keras functional API the reference's hot path uses, re-hosted on libty1.so.

The reference builds its model symbolically (Input -> Conv2D -> BatchNormalization -> LeakyReLU ->
MaxPooling2D ... -> Flatten -> Dense -> ReshapeYOLO) and then calls Model(inputs, outputs) with
.load_weights / .predict / .compile / .fit_generator / .save_weights
(/root/reference/tiny_yolov1.py:25-31, /root/reference/train.py:22-25,59-60,94-105).  Here the
layer objects only RECORD the graph; Model pattern-matches it into fused gfx950 kernels
(conv+BN+leaky[+pool], split-K Dense, head activation) -- nothing runs per layer and nothing runs
on the CPU.  Graphs outside that pattern are rejected loudly.
"""
import numpy as np
import torch

from . import params as P


class SymbolicTensor:
    def __init__(self, shape, layer=None, parent=None):
        self.shape = tuple(shape)          # (None, H, W, C): callers read inputs.shape[1:]
        self.layer, self.parent = layer, parent


def Input(shape):
    return SymbolicTensor((None,) + tuple(shape))


def l2(weight):
    """kernel_regularizer=l2(5e-4) of the reference: constant w.r.t. the only trainable layer (the
    conv layers are frozen), so it never affects a gradient; accepted and ignored."""
    return ('l2', float(weight))


class Layer:
    def __init__(self, name=None, trainable=True, **kwargs):
        self.name, self.trainable = name, trainable

    def compute_output_shape(self, input_shape):
        return input_shape

    def __call__(self, x):
        return SymbolicTensor(self.compute_output_shape(x.shape), self, x)

    def get_config(self):
        return {'name': self.name, 'trainable': self.trainable}


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, padding='valid', use_bias=True, kernel_regularizer=None, **kw):
        super().__init__(**kw)
        self.filters, self.kernel_size, self.padding, self.use_bias = filters, tuple(kernel_size), padding, use_bias

    def compute_output_shape(self, s):
        return s[:3] + (self.filters,)


class BatchNormalization(Layer):
    epsilon = P.BN_EPS            # Keras default 1e-3


class LeakyReLU(Layer):
    def __init__(self, alpha=0.3, **kw):
        super().__init__(**kw)
        self.alpha = alpha


class MaxPooling2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding='valid', **kw):
        super().__init__(**kw)
        self.pool_size, self.strides, self.padding = tuple(pool_size), tuple(strides or pool_size), padding

    def compute_output_shape(self, s):
        return (s[0], -(-s[1] // 2), -(-s[2] // 2), s[3])


class Flatten(Layer):
    def compute_output_shape(self, s):
        return (s[0], int(np.prod(s[1:])))


class Dense(Layer):
    def __init__(self, units, activation=None, **kw):
        super().__init__(**kw)
        self.units, self.activation = units, activation

    def compute_output_shape(self, s):
        return (s[0], self.units)


class Adam:
    """tensorflow.keras.optimizers.Adam(lr=0.001) with Keras defaults (train.py:59)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, learning_rate=None):
        self.lr = learning_rate if learning_rate is not None else lr
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon


def _chain(outputs):
    layers, t = [], outputs
    while t.layer is not None:
        layers.append(t.layer)
        t = t.parent
    return t, layers[::-1]


class Model:
    """Model(inputs=..., outputs=...) for the tiny-YOLOv1 pattern."""

    def __init__(self, inputs, outputs, device='cuda:0', seed=1, dtype='f32'):
        """dtype='bf16' runs inference with bf16 conv activations/weights (engine.DeviceModel); training needs 'f32'."""
        root, layers = _chain(outputs)
        if root is not inputs:
            raise ValueError('outputs are not derived from inputs')
        self.inputs, self.outputs, self.layers = inputs, outputs, layers
        self.device, self.dtype = device, dtype
        self._parse(layers)
        self.params = self._init_params(seed)
        self._engine = None
        self._trainer = None
        self._trained = False          # the engine's Dense weights are ahead of self.params (set by train_on_batch)
        self.loss, self.optimizer = None, None

    # ------------------------------------------------------------------ graph -> fused plan
    def _parse(self, layers):
        from .models.model_tiny_yolov1 import ReshapeYOLO
        H, W, cin = self.inputs.shape[1:]
        if H != W:
            raise ValueError('square inputs only')
        self.img_size = H
        specs, names, i = [], [], 0
        while i < len(layers) and isinstance(layers[i], Conv2D):
            conv = layers[i]
            if conv.kernel_size != (3, 3) or conv.padding != 'same' or conv.use_bias:
                raise ValueError(f'{conv.name}: only Conv2D(3x3, padding="same", use_bias=False) is accelerated')
            if not (i + 2 < len(layers) and isinstance(layers[i + 1], BatchNormalization) and isinstance(layers[i + 2], LeakyReLU)):
                raise ValueError(f'{conv.name}: expected Conv2D -> BatchNormalization -> LeakyReLU')
            if abs(layers[i + 2].alpha - P.LEAKY_ALPHA) > 1e-12:
                raise ValueError('LeakyReLU alpha must be 0.1')
            if conv.trainable or layers[i + 1].trainable:
                raise ValueError(f'{conv.name}: the backbone must be frozen (trainable=False), as in the reference')
            bn_name = layers[i + 1].name
            i += 3
            pool = i < len(layers) and isinstance(layers[i], MaxPooling2D)
            if pool:
                mp = layers[i]
                if mp.pool_size != (2, 2) or mp.strides != (2, 2):
                    raise ValueError('only MaxPooling2D((2,2), strides=(2,2)) is accelerated')
                i += 1
            specs.append((cin, conv.filters, pool))
            names.append((conv.name, bn_name))
            cin = conv.filters
        tail = layers[i:]
        if not (len(tail) == 3 and isinstance(tail[0], Flatten) and isinstance(tail[1], Dense)
                and isinstance(tail[2], ReshapeYOLO)):
            raise ValueError('expected ... -> Flatten -> Dense -> ReshapeYOLO after the conv blocks')
        if tail[1].activation not in (None, 'linear'):
            raise ValueError('Dense must be linear')
        self.conv_specs, self.conv_names = specs, names
        self.dense_name = tail[1].name
        self.head = tail[2]
        S0, S1, D = self.head.target_shape
        self.S, self.B, self.C = S0, self.head.B, self.head.C
        if S0 != S1 or D != self.C + 5 * self.B or tail[1].units != S0 * S1 * D:
            raise ValueError('Dense units / ReshapeYOLO target_shape mismatch')
        self.features = int(np.prod(self._feature_shape(specs)[1:]))

    def _feature_shape(self, specs):
        h = self.img_size
        for _, _, pool in specs:
            h = h // 2 if pool else h
        return (None, h, h, specs[-1][1])

    def _init_params(self, seed):
        """Keras initialisers: glorot_uniform kernels, zero biases, identity BatchNormalization."""
        rng = np.random.default_rng(seed)
        p = {}
        for i, (cin, cout, _) in enumerate(self.conv_specs):
            p[f'conv{i}_w'] = P.glorot_uniform(rng, (3, 3, cin, cout), 9 * cin, 9 * cout)
            p[f'bn{i}_gamma'], p[f'bn{i}_beta'] = np.ones(cout, np.float32), np.zeros(cout, np.float32)
            p[f'bn{i}_mean'], p[f'bn{i}_var'] = np.zeros(cout, np.float32), np.ones(cout, np.float32)
        units = self.S * self.S * (self.C + 5 * self.B)
        p['dense_w'] = P.glorot_uniform(rng, (self.features, units), self.features, units)
        p['dense_b'] = np.zeros(units, np.float32)
        return p

    # ------------------------------------------------------------------ weights
    def set_params(self, params):
        self.params = {k: np.ascontiguousarray(v, np.float32) for k, v in params.items()}
        self._engine = self._trainer = None
        self._trained = False

    def _keras_names(self):
        """Keras weight names -> our parameter keys (for by_name loading)."""
        m = {}
        for i, (cn, bn) in enumerate(self.conv_names):
            m[f'{cn}/kernel'] = f'conv{i}_w'
            for k in ('gamma', 'beta'):
                m[f'{bn}/{k}'] = f'bn{i}_{k}'
            m[f'{bn}/moving_mean'], m[f'{bn}/moving_variance'] = f'bn{i}_mean', f'bn{i}_var'
        m[f'{self.dense_name}/kernel'], m[f'{self.dense_name}/bias'] = 'dense_w', 'dense_b'
        return m

    def load_weights(self, path, by_name=True):
        """.npz written by save_weights (keys = Keras weight names).  As in Keras' by_name=True,
        only names that exist in this model are loaded and the rest keep their initial values (so
        the reference's converter names `convolutional_N` match nothing but connected_0: SURVEY Q11).
        HDF5 needs h5py, which this image lacks: convert once with yolo_amd.darknet or save as .npz."""
        if str(path).endswith(('.hdf5', '.h5')):
            try:
                import h5py  # noqa: F401
            except ImportError as e:
                raise ImportError('HDF5 weights need h5py (absent here); use a .npz or a Darknet .weights file '
                                  '(yolo_amd.darknet.load_darknet_weights)') from e
            raise NotImplementedError('HDF5 reader not wired; use .npz / Darknet .weights')
        self._pull_trained()             # by_name may load only part of the model: keep what training changed
        names = self._keras_names()
        loaded = []
        with np.load(path, allow_pickle=False) as z:
            for k in z.files:
                if k in names and self.params[names[k]].shape == z[k].shape:
                    self.params[names[k]] = np.ascontiguousarray(z[k], np.float32)
                    loaded.append(k)
        self._engine = self._trainer = None
        self._trained = False
        return loaded

    def save_weights(self, path):
        self._pull_trained()
        inv = {v: k for k, v in self._keras_names().items()}
        np.savez(path, **{inv[k]: v for k, v in self.params.items()})

    def _pull_trained(self):
        """Copy the trained Dense kernel / bias back from the device.  Keyed on "a step has run since the last pull",
        not on the trainer object: compile() drops the trainer (new optimizer state) but the engine keeps the weights."""
        if self._trained and self._engine is not None:
            eng = self._engine
            self._trained = False
            self.params['dense_w'] = P.dense_from_device_layout(eng.dense_wt.cpu().numpy(), eng.units)
            self.params['dense_b'] = eng.dense_b.cpu().numpy()

    # ------------------------------------------------------------------ execution
    @property
    def engine(self):
        if self._engine is None:
            from .engine import DeviceModel
            self._engine = DeviceModel(self.params, self.S, self.B, self.C, self.img_size, self.device, self.conv_specs,
                                       dtype=self.dtype)
        return self._engine

    def _to_device(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, np.float32))     # load_img yields float64: cast as Keras does
        return x.to(self.device, torch.float32).contiguous()

    def predict(self, x, batch_size=32, as_numpy=True):
        """-> (N,S,S,C+5B) float32, np.ndarray like Keras (or a device tensor with as_numpy=False)."""
        x = self._to_device(x)
        outs = [self.engine.predict(x[i:i + batch_size]).clone() for i in range(0, x.shape[0], batch_size)]
        y = torch.cat(outs, 0) if len(outs) != 1 else outs[0]
        return y.cpu().numpy() if as_numpy else y

    def compile(self, loss=None, optimizer='adam', **kw):
        self.loss = loss
        self.optimizer = Adam() if isinstance(optimizer, str) else optimizer
        self._trainer = None

    def _get_trainer(self, group=None):
        if self._trainer is None:
            from .engine import Trainer
            o = self.optimizer or Adam()
            self._trainer = Trainer(self.engine, o.lr, o.beta_1, o.beta_2, o.epsilon, group)
        return self._trainer

    def train_on_batch(self, x, y):
        """One optimizer step on (x, y_true); returns the batch-SUM loss as a float."""
        t = self._get_trainer()
        self._trained = True
        return float(t.step(self._to_device(x), self._to_device(y)).item())

    def test_on_batch(self, x, y):
        from . import ops
        yp = self.predict(x, batch_size=len(x), as_numpy=False)
        return float(ops.yolo_loss_fwd_bwd(self._to_device(y), yp, self.B, self.C, self.img_size, want_grad=False)[0].item())

    LOOKAHEAD = 2      # batches of a generator that share one pass of the frozen backbone (engine.Trainer.steps)

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, validation_data=None, validation_steps=None,
                      callbacks=None, verbose=1, **kw):
        """Minimal epoch loop over a Sequence-like generator (len(), [i] -> (X, y)), /root/reference/train.py:94-104; the
        Keras callback machinery (ModelCheckpoint / EarlyStopping) is out of scope.
        Consecutive batches of equal shape go through Trainer.steps LOOKAHEAD at a time (one pass of the frozen backbone,
        then one optimisation step per batch, in order), the update stays in flight under the next pass, and the losses
        stay on the device until the epoch ends: no host synchronisation per step."""
        history = {'loss': [], 'val_loss': []}
        steps = steps_per_epoch or len(generator)
        t = self._get_trainer()
        for ep in range(epochs):
            losses, i = [], 0
            while i < steps:
                group = [generator[j] for j in range(i, min(i + self.LOOKAHEAD, steps))]
                while len(group) > 1 and any(np.shape(g[0]) != np.shape(group[0][0]) for g in group):
                    group.pop()                      # a ragged last batch takes its step alone
                X = torch.cat([self._to_device(g[0]) for g in group]) if len(group) > 1 else self._to_device(group[0][0])
                y = torch.cat([self._to_device(g[1]) for g in group]) if len(group) > 1 else self._to_device(group[0][1])
                self._trained = True
                losses.append(t.steps(X, y, X.shape[0] // len(group), wait=False))
                i += len(group)
            t.join()
            history['loss'].append(float(torch.cat(losses).sum().item()) / steps)
            if validation_data is not None:
                vs = validation_steps or len(validation_data)
                history['val_loss'].append(sum(self.test_on_batch(*validation_data[i]) for i in range(vs)) / vs)
            if verbose:
                print(f'epoch {ep + 1}/{epochs} loss {history["loss"][-1]:.4f}')
            if hasattr(generator, 'on_epoch_end'):
                generator.on_epoch_end()
        return history
