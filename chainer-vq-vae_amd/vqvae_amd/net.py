"""Encoder / ConditionEmbed / VAE of the VQ-VAE audio model.

Public surface (class names, constructor and call signatures, sub-link names and
therefore parameter paths) is the reference's net.py (Encoder net.py:8-26,
ConditionEmbed net.py:29-64, VAE net.py:67-96); the bodies are written for this
runtime: activations fused into conv epilogues, the condition tensor kept at the
latent rate until the decoder consumes it, one nearest-code search per step.
"""
from . import backend, core, functions as F, links as L, prepack
from .core import Chain, Variable
from .utils import VQ

# (ksize, stride, pad) of every encoder stage: 4-tap, stride-2, pad-1 along time (net.py:12-17)
_DOWN = dict(ksize=(4, 1), stride=(2, 1), pad=(1, 0))
# dilations of the "same"-padded 3-tap condition convs (net.py:34-43)
_COND_DILATIONS = (1, 2, 4, 8, 16)


class Encoder(Chain):
    """Six stride-2 convs, 64x temporal down-sampling; ReLU after all but the last."""

    depth = 6

    def __init__(self, d):
        super(Encoder, self).__init__()
        with self.init_scope():
            width_in = 1
            for stage in range(1, self.depth + 1):
                setattr(self, 'conv%d' % stage, L.Convolution2D(width_in, d, **_DOWN))
                width_in = d

    def __call__(self, x):
        h = x
        for stage in range(1, self.depth + 1):
            conv = getattr(self, 'conv%d' % stage)
            h = conv(h, relu=(stage < self.depth))      # ReLU fused in the GEMM epilogue
        return h


class ConditionEmbed(Chain):
    """Local (latent) condition: five dilated 3-tap convs + ReLU, then x`upscale_factor`
    align-corners up-sampling; global condition: speaker embedding broadcast over time;
    both concatenated on the channel axis."""

    def __init__(self, n_global_cond, global_embed_dim, local_embed_dim, upscale_factor=64):
        super(ConditionEmbed, self).__init__()
        with self.init_scope():
            for i, dil in enumerate(_COND_DILATIONS, start=1):
                # in_channels=None: inferred from the first input, like the reference
                setattr(self, 'local_embed%d' % i, L.DilatedConvolution2D(
                    None, local_embed_dim, (3, 1), pad=(dil, 0), dilate=(dil, 1)))
            self.global_embed = L.EmbedID(n_global_cond, global_embed_dim)
        self.upscale_factor = upscale_factor

    def __call__(self, local_condition, global_condition):
        convs = [getattr(self, 'local_embed%d' % i) for i in range(1, len(_COND_DILATIONS) + 1)]
        h = local_condition
        width = h.shape[1]
        for c in convs:                        # lazily shaped (net.py:34-43): the first call creates the parameters, as the links would
            if c.W.data is None:
                c._initialize_params(width)
                if isinstance(h.data, backend.DeviceArray):
                    c.W.to_gpu()
            width = c.W.shape[0]
        h = F.conv_stack(h, convs)             # one launch per direction where the library serves the shape (csrc/latent.hip)
        # resize_images(local) ++ resize_images(EmbedID(speaker)) ++ concat in one node that
        # keeps the result at the latent rate until a consumer needs the full-rate tensor
        return F.condition_assemble(h, self.global_embed.W, global_condition, self.upscale_factor)


class VAE(Chain):
    """encoder -> VQ (straight-through) -> condition embed -> decoder, three losses:
    loss1 reconstruction, loss2 codebook, loss3 = beta * commitment."""

    def __init__(self, encoder, decoder, condition_embed, d, k, beta, loss_func):
        super(VAE, self).__init__()
        self.beta = beta
        self.loss_func = loss_func
        with self.init_scope():
            self.encoder = encoder
            self.vq = VQ(k, d)
            self.condition_embed = condition_embed
            self.decoder = decoder

    def __call__(self, x_enc, x_dec, global_condition, t):
        # the weight slabs of the step's small convs (encoder, condition embed, proj1 / proj2; both directions), as the
        # previous step used them: packed on the side stream now, in a few batched launches, instead of one pack in front
        # of every conv on the critical path (prepack.py)
        if isinstance(x_dec, backend.DeviceArray):
            prepack.prefetch()
        # the decoder's weight slabs are packed on the side stream while the encoder / quantiser / condition embed run
        dec = self.decoder
        dec = getattr(dec, 'target' if core.config.train else 'ema', dec)
        rn = getattr(dec, 'resnet', None)
        # (issued BEHIND the encoder's launches -- see prepack_async -- but ordered behind the point where the step started)
        start = backend.Event().record(backend.stream()) if isinstance(x_dec, backend.DeviceArray) else None
        z = self.encoder(x_enc)
        if hasattr(rn, 'prepack_async') and isinstance(x_dec, backend.DeviceArray):
            rn.prepack_async(x_dec.shape[0], x_dec.shape[2] if x_dec.ndim == 4 else x_dec.shape[1], after=start)
        z_const = Variable(z.data)            # stop-gradient view of the latents

        # Two quantiser applications route the gradients (net.py:82-83): through `e` the
        # reconstruction loss reaches the encoder (identity backward) and, discarded by the
        # updater, the codebook; through `e_cb` only the codebook is reached.  The second
        # application reuses the first one's nearest-code search (same data, same codebook).
        e = self.vq(z)
        e_cb = self.vq(z_const)

        y = self.decoder(x_dec, self.condition_embed(e, global_condition))

        loss1 = self.loss_func(y, t)
        loss2 = F.mean_squared_difference(z_const, e_cb)                    # F.mean((z.data - e_) ** 2), net.py:90
        loss3 = self.beta * F.mean_squared_difference(z, Variable(e.data))  # beta * F.mean((z - e.data) ** 2), net.py:91
        core.report({'loss1': loss1, 'loss2': loss2, 'loss3': loss3,
                     'loss': loss1 + loss2 + loss3}, self)
        prepack.join()
        return loss1, loss2, loss3
