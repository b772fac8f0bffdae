"""Darknet .cfg + .weights -> parameter dict, loaded by POSITION straight into the HIP model
(SURVEY 8f row 1).  Byte layout as /root/reference/convert.py reads it:
  header          4 x int32                                              (convert.py:59-60)
  [convolutional] bias/beta[F]; if batch_normalize: gamma[F], mean[F], var[F]; weights (F,Cin,k,k)
                  -> transposed to (k,k,Cin,F)                            (convert.py:110-144)
  [connected]     bias[out]; weights (out,in) -> transposed to (in,out)   (convert.py:211-229)
Like the reference, the FC weights are NOT permuted from Darknet's CHW flatten to Keras' HWC flatten
(SURVEY Q10): we follow the reference, we do not "correct" it.
"""
import configparser
import io
from collections import defaultdict

import numpy as np


def parse_cfg(path):
    """Darknet cfg sections in order, as (type, dict) pairs (duplicate section names allowed)."""
    counts, out = defaultdict(int), io.StringIO()
    with open(path) as f:
        for line in f:
            if line.startswith('['):
                name = line.strip().strip('[]')
                line = f'[{name}_{counts[name]}]\n'
                counts[name] += 1
            out.write(line)
    out.seek(0)
    cp = configparser.ConfigParser()
    cp.read_file(out)
    return [(s.rsplit('_', 1)[0], dict(cp[s])) for s in cp.sections()]


def conv_specs_from_cfg(sections):
    """-> (img_size, [(cin, cout, pool)], fc_units) for a conv/maxpool/connected stack."""
    net = dict(sections[0][1]) if sections[0][0] == 'net' else {}
    crop = next((d for t, d in sections if t == 'crop'), None)
    size = int(crop['crop_height']) if crop else int(net['height'])
    cin, specs, units = int(net.get('channels', 3)), [], None
    for t, d in sections:
        if t == 'convolutional':
            if int(d['size']) != 3 or int(d['stride']) != 1 or int(d['pad']) != 1 or d['activation'] != 'leaky' \
                    or 'batch_normalize' not in d:
                raise ValueError('only 3x3/1 pad=1 leaky batch-normalised convolutions are supported')
            specs.append([cin, int(d['filters']), False])
            cin = int(d['filters'])
        elif t == 'maxpool':
            if int(d['size']) != 2 or int(d['stride']) != 2 or not specs or specs[-1][2]:
                raise ValueError('only one 2x2/2 maxpool directly after a convolution is supported')
            specs[-1][2] = True
        elif t == 'connected':
            units = int(d['output'])
    return size, [tuple(s) for s in specs], units


def write_cfg(path, specs, img_size=448, units=1470, S=7, B=2, C=20):
    """Write a Darknet cfg for a conv/maxpool/connected/detection stack: the section pattern of
    /root/reference/cfg/yolov1-tiny.cfg generated from (cin, cout, pool) specs (the inverse of conv_specs_from_cfg;
    training hyper-parameters of [net] are not part of the model and are left out)."""
    out = [f'[net]\nbatch=1\nsubdivisions=1\nheight={img_size}\nwidth={img_size}\nchannels={specs[0][0]}\n']
    for _, cout, pool in specs:
        out.append(f'[convolutional]\nbatch_normalize=1\nfilters={cout}\nsize=3\nstride=1\npad=1\nactivation=leaky\n')
        if pool:
            out.append('[maxpool]\nsize=2\nstride=2\n')
    out.append(f'[connected]\noutput= {units}\nactivation=linear\n')
    out.append(f'[detection]\nclasses={C}\ncoords=4\nrescore=1\nside={S}\nnum={B}\nsoftmax=0\nsqrt=1\n')
    with open(path, 'w') as f:
        f.write('\n'.join(out))


def load_darknet_weights(cfg_path, weights_path):
    """-> (params dict in Keras layouts, img_size, conv_specs)."""
    size, specs, units = conv_specs_from_cfg(parse_cfg(cfg_path))
    blob = np.fromfile(weights_path, dtype=np.uint8)
    header = blob[:16].view(np.int32)
    data = blob[16:].view(np.float32)
    pos, p, h = 0, {'header': header.copy()}, size

    def take(n):
        nonlocal pos
        if pos + n > data.size:
            raise ValueError(f'{weights_path}: truncated (need {pos + n} floats, have {data.size})')
        out = data[pos:pos + n]
        pos += n
        return out

    for i, (cin, cout, pool) in enumerate(specs):
        p[f'bn{i}_beta'] = take(cout).copy()
        bn = take(3 * cout).reshape(3, cout)
        p[f'bn{i}_gamma'], p[f'bn{i}_mean'], p[f'bn{i}_var'] = bn[0].copy(), bn[1].copy(), bn[2].copy()
        w = take(cout * cin * 9).reshape(cout, cin, 3, 3)
        p[f'conv{i}_w'] = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))
        h = h // 2 if pool else h
    if units is not None:
        feats = h * h * specs[-1][1]
        p['dense_b'] = take(units).copy()
        p['dense_w'] = np.ascontiguousarray(take(units * feats).reshape(units, feats).T)
    p['floats_read'] = pos
    return p, size, specs


def write_darknet_weights(path, params, specs, header=(0, 1, 0, 0)):
    """Inverse of load_darknet_weights (used by the tests to build a synthetic file)."""
    with open(path, 'wb') as f:
        f.write(np.asarray(header, np.int32).tobytes())
        for i, (cin, cout, pool) in enumerate(specs):
            f.write(params[f'bn{i}_beta'].astype(np.float32).tobytes())
            f.write(np.stack([params[f'bn{i}_gamma'], params[f'bn{i}_mean'], params[f'bn{i}_var']]).astype(np.float32).tobytes())
            f.write(np.ascontiguousarray(np.transpose(params[f'conv{i}_w'], (3, 2, 0, 1))).astype(np.float32).tobytes())
        f.write(params['dense_b'].astype(np.float32).tobytes())
        f.write(np.ascontiguousarray(params['dense_w'].T).astype(np.float32).tobytes())
