"""Encoder / ConditionEmbed / VAE -- mirrors the reference's net.py (Encoder
net.py:8-26, ConditionEmbed 29-64, VAE 67-96), same constructor and call
signatures, same three returned losses."""
from . import core, functions as F, links as L
from .core import Chain, Variable
from .utils import VQ


class Encoder(Chain):
    def __init__(self, d):
        super(Encoder, self).__init__()
        with self.init_scope():
            self.conv1 = L.Convolution2D(1, d, (4, 1), (2, 1), (1, 0))
            self.conv2 = L.Convolution2D(d, d, (4, 1), (2, 1), (1, 0))
            self.conv3 = L.Convolution2D(d, d, (4, 1), (2, 1), (1, 0))
            self.conv4 = L.Convolution2D(d, d, (4, 1), (2, 1), (1, 0))
            self.conv5 = L.Convolution2D(d, d, (4, 1), (2, 1), (1, 0))
            self.conv6 = L.Convolution2D(d, d, (4, 1), (2, 1), (1, 0))

    def __call__(self, x):
        # F.relu(conv(x)) with the ReLU fused into the conv epilogue (net.py:20-24)
        h = self.conv1(x, relu=True)
        h = self.conv2(h, relu=True)
        h = self.conv3(h, relu=True)
        h = self.conv4(h, relu=True)
        h = self.conv5(h, relu=True)
        z = self.conv6(h)
        return z


class ConditionEmbed(Chain):
    def __init__(self, n_global_cond, global_embed_dim, local_embed_dim, upscale_factor=64):
        super(ConditionEmbed, self).__init__()
        with self.init_scope():
            self.local_embed1 = L.DilatedConvolution2D(
                None, local_embed_dim, (3, 1), pad=(1, 0), dilate=(1, 1))
            self.local_embed2 = L.DilatedConvolution2D(
                None, local_embed_dim, (3, 1), pad=(2, 0), dilate=(2, 1))
            self.local_embed3 = L.DilatedConvolution2D(
                None, local_embed_dim, (3, 1), pad=(4, 0), dilate=(4, 1))
            self.local_embed4 = L.DilatedConvolution2D(
                None, local_embed_dim, (3, 1), pad=(8, 0), dilate=(8, 1))
            self.local_embed5 = L.DilatedConvolution2D(
                None, local_embed_dim, (3, 1), pad=(16, 0), dilate=(16, 1))
            self.global_embed = L.EmbedID(n_global_cond, global_embed_dim)
        self.upscale_factor = upscale_factor

    def __call__(self, local_condition, global_condition):
        local_condition = self.local_embed1(local_condition, relu=True)
        local_condition = self.local_embed2(local_condition, relu=True)
        local_condition = self.local_embed3(local_condition, relu=True)
        local_condition = self.local_embed4(local_condition, relu=True)
        local_condition = self.local_embed5(local_condition, relu=True)
        # resize_images(local) ++ resize_images(EmbedID(speaker)) ++ concat (net.py:54-63),
        # written straight into the concatenated tensor
        condition = F.condition_assemble(local_condition, self.global_embed.W, global_condition,
                                         self.upscale_factor)
        return condition


class VAE(Chain):
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
        # forward
        z = self.encoder(x_enc)
        e = self.vq(z)
        e_ = self.vq(Variable(z.data))
        local_condition = e
        condition = self.condition_embed(local_condition, global_condition)
        y = self.decoder(x_dec, condition)

        # calculate loss
        loss1 = self.loss_func(y, t)
        loss2 = F.mean((Variable(z.data) - e_) ** 2)
        loss3 = self.beta * F.mean((z - Variable(e.data)) ** 2)
        loss = loss1 + loss2 + loss3
        core.report(
            {'loss1': loss1, 'loss2': loss2, 'loss3': loss3, 'loss': loss}, self)
        return loss1, loss2, loss3
