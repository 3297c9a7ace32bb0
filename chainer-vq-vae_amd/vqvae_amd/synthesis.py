"""The forward half of generate.py as a library call: encode a waveform, quantise it, embed the
condition (optionally for another speaker -- voice conversion, generate.py:19-21, 94-98) and run
the autoregressive decoder on the device.

generate.py's file handling (librosa load/trim, snapshot paths, write_wav) stays outside the hot
path; `serializers.load_npz(path, link, prefix)` loads the reference's snapshot layout
(generate.py:67-81) into the links handed to this function.
"""
import numpy as np

from . import backend, core
from .core import Variable
from .utils import MuLaw


def synthesize(encoder, vq, decoder, condition_embed, x_enc, global_condition, rng=None,
               quantize=256, use_logistic=False, n_steps=None, persistent=True):
    """generate.py:100-148.

    encoder / vq / decoder / condition_embed: device-resident links (decoder = the WaveNet, EMA or
        target copy as generate.py:70-79 chooses).
    x_enc: host or device float32 (n, 1, T+1, 1) normalised waveform(s) (utils.py:99-100; the
        reference runs n = 1).   global_condition: int32 (n,) speaker ids.
    rng: numpy.random.RandomState (default: a fresh RandomState(), the stand-in for NumPy's global
        RNG that generate.py:117 / 136 draws from).  The uniforms are drawn up front, one per step
        (softmax output) or nr_mix per step (mixture of logistics), in the reference's order.
    Returns (wave, output): wave float64/float32 host array (n, T) as written by generate.py:146-149
        (mu-law expanded for the softmax output), output = the raw sampled bins / values."""
    rng = np.random.RandomState() if rng is None else rng
    xd = x_enc if backend.is_device(x_enc) else backend.to_device(np.asarray(x_enc, np.float32))
    gd = global_condition if backend.is_device(global_condition) else \
        backend.to_device(np.asarray(global_condition, np.int32))
    n = xd.shape[0]
    with core.using_config('train', False), core.no_backprop_mode():
        z = encoder(Variable(xd))                                   # generate.py:95
        e = vq(z)                                                   # generate.py:96
        condition = condition_embed(e, Variable(gd))                # generate.py:97-98
    cond = condition.data
    T = cond.shape[2]
    steps = T - 1 if n_steps is None else n_steps                   # generate.py:105
    nr_mix = decoder.proj2.W.shape[0] // 3
    if use_logistic:
        u = rng.uniform(0, 1, (max(steps, 1), n, nr_mix))            # generate.py:117
    else:
        u = rng.random_sample((max(steps, 1), n))                    # inside numpy.random.choice, generate.py:136
    out = decoder.generate_sequence(cond, u, n_steps=steps, persistent=persistent).get()
    if use_logistic:
        return out, out                                             # generate.py:147
    return MuLaw(quantize).itransform(out), out                     # generate.py:149
