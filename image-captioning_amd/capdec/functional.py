"""Module-level compositions of the HIP ops used by the ``models`` classes.

Inference compositions call the forward kernels directly; training compositions (autograd enabled and
something requires grad) go through ``capdec.autograd`` Functions whose forward and backward are both
HIP kernels.  Nothing here falls back to ATen arithmetic.
"""
import torch

from . import ops


def _needs_graph(module, *tensors):
    return torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or
                                        any(p.requires_grad for p in module.parameters()))


def linear_dropout_bn(model, parts, x, dropout_first=False):
    """Feature head: for each (fc, lo, hi) in ``parts`` project x[:, lo:hi] with fc, concatenate the
    results column-wise, apply ``model.dropout`` (before the projection when ``dropout_first``) and
    ``model.batchnorm``.  Returns [B, len(parts)*E].

    eval: dropout is the identity and the BatchNorm (running statistics) is folded into the contraction's
    epilogue, so the head is one kernel per projection.  train: see ``capdec.autograd``.
    """
    bn = model.batchnorm
    if model.training:
        from . import autograd as AG
        return AG.head_train(model, parts, x, dropout_first)
    B = x.shape[0]
    E = parts[0][0].weight.shape[0]
    # a long contraction with few output tiles (the 86436-wide YOLO projection) is split over workgroups, its partial
    # sums added in a fixed order: the memory rows, and with them the decoded tokens, repeat exactly from run to run
    # (likewise a small batch in front of a 1000- or 2048-wide feature: at batch 32 the CNN head is four 64x64 tiles
    #  that each walk all 2048 k, 26.7 us on four of the chip's 256 compute units)
    few_tiles = ((B + 63) // 64) * ((E + 63) // 64) <= 32
    return ops.head_eval(x, [(fc.weight, fc.bias, lo, hi) for fc, lo, hi in parts], bn.weight, bn.bias,
                         bn.running_mean, bn.running_var, bn.eps,
                         lambda K: K >= 8192 or (few_tiles and K >= 1024))


def decoder_forward(dec, enc_output, captions, take_last=False):
    """Autograd-connected teacher-forced forward (reference transformer.py:125-138)."""
    from . import autograd as AG
    return AG.decoder_forward(dec, enc_output, captions, take_last)


def decoder_rnn_forward(rnn, features, captions):
    """``DecoderRNN.forward`` (reference cnnrnn_model.py:20-34): logits [B,T,V]."""
    if _needs_graph(rnn, features):
        from . import autograd as AG
        return AG.decoder_rnn_forward(rnn, features, captions)
    x0 = rnn.image_input(features)
    w = (rnn.lstm.weight_ih_l0, rnn.lstm.weight_hh_l0, rnn.lstm.bias_ih_l0, rnn.lstm.bias_hh_l0)
    h, c = ops.lstm_step(x0, None, None, *w)
    B, T = captions.shape
    emb = ops.embed_pe(captions, rnn.embed.weight, None).view(B, T, -1)
    Hd = h.shape[1]
    hs = torch.empty((B, T, Hd), device=h.device, dtype=torch.float32)
    for t in range(T):
        h, c = ops.lstm_step(emb[:, t], h, c, *w, h_out=hs[:, t])   # h' lands in the [B,T,Hd] slab
    return ops.linear(hs.view(B * T, Hd), rnn.fc2.weight, rnn.fc2.bias).view(B, T, -1)
