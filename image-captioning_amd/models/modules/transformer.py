"""Post-LN Transformer caption decoder on hand-written gfx950 kernels.

Mirrors the class and parameter names of the reference's ``models/modules/transformer.py`` so that
``state_dict`` keys match (``decoder_layers.{l}.self_attn.W_q.weight`` ...).  The ``torch.nn`` layers
below are parameter containers only: no ``nn.Linear.forward`` / ``nn.LayerNorm.forward`` is ever
called -- all arithmetic runs in libcapdec.so.  There is no CPU path.
"""
import math
from abc import ABC, abstractmethod

import torch
import torch.nn as nn

from capdec import engine
from capdec import functional as CF


class MultiHeadAttention(nn.Module):
    """Parameter holder for W_q/W_k/W_v/W_o (reference transformer.py:7-19)."""

    def __init__(self, d_model, num_heads):
        super().__init__()
        assert d_model % num_heads == 0
        self.d_model, self.num_heads, self.d_k = d_model, num_heads, d_model // num_heads
        self.W_q = nn.Linear(d_model, d_model)
        self.W_k = nn.Linear(d_model, d_model)
        self.W_v = nn.Linear(d_model, d_model)
        self.W_o = nn.Linear(d_model, d_model)


class PositionWiseFeedForward(nn.Module):
    """Parameter holder for fc1/fc2 (reference transformer.py:46-51)."""

    def __init__(self, d_model, d_ff):
        super().__init__()
        self.fc1 = nn.Linear(d_model, d_ff)
        self.fc2 = nn.Linear(d_ff, d_model)


class PositionalEncoding(nn.Module):
    """Sinusoid table registered as buffer ``pe`` [1, max_len, E] (reference transformer.py:56-67)."""

    def __init__(self, d_model, max_seq_length):
        super().__init__()
        pos = torch.arange(max_seq_length, dtype=torch.float32).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        table = torch.zeros(max_seq_length, d_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table.unsqueeze(0))


class DecoderLayer(nn.Module):
    """Parameter holder for one post-LN layer (reference transformer.py:72-81)."""

    def __init__(self, d_model, num_heads, d_ff, dropout):
        super().__init__()
        self.self_attn = MultiHeadAttention(d_model, num_heads)
        self.cross_attn = MultiHeadAttention(d_model, num_heads)
        self.feed_forward = PositionWiseFeedForward(d_model, d_ff)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)


class DecoderTransformer(ABC, nn.Module):
    """Same constructor, parameter names and public methods as the reference class
    (transformer.py:92-175); ``decoder_forward`` and ``greedy_inference`` run on the HIP kernels."""

    def __init__(self, embed_size, vocab_size, num_heads, num_layers, dropout, max_seq_length):
        super().__init__()
        self.num_heads = num_heads
        self.dropout = nn.Dropout(dropout)
        self.decoder_layers = nn.ModuleList(
            [DecoderLayer(embed_size, num_heads, embed_size, dropout) for _ in range(num_layers)])
        self.positional_encoding = PositionalEncoding(embed_size, max_seq_length)
        self.fc2 = nn.Linear(embed_size, vocab_size)
        self.decoder_embedding = nn.Embedding(vocab_size, embed_size)

    @abstractmethod
    def precompute_image(self, images):
        pass

    @abstractmethod
    def forward(self, *args, **kwargs):
        pass

    @abstractmethod
    def caption_images(self, *args, **kwargs):
        pass

    @abstractmethod
    def caption_images_beam_search(self, *args, **kwargs):
        pass

    # -- teacher-forced forward ----------------------------------------------------------------
    def decoder_forward(self, enc_output, captions, take_last=False):
        """enc_output [B,S,E] fp32, captions [B,T] int64 -> logits [B,T,V] ([B,V] with take_last).

        With autograd enabled and any input/parameter requiring grad, the graph is built from
        ``capdec.functional`` ops (HIP forward and backward kernels); otherwise one C call runs the
        whole stack."""
        needs_graph = torch.is_grad_enabled() and (
            enc_output.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_graph or (self.training and self.dropout.p > 0 and not take_last):
            return CF.decoder_forward(self, enc_output, captions, take_last)
        return engine.tf_forward_logits(self, enc_output, captions, take_last)

    # -- greedy decode ---------------------------------------------------------------------------
    def greedy_ids(self, enc_output, sos, eos, max_length, **kw):
        """Token ids instead of strings: (ids int64 [B,max_length] padded with -1, lengths [B]) on the CPU."""
        ids, lengths, _ = engine.tf_greedy_ids(self, enc_output, max_length, sos, eos, **kw)
        return ids, lengths

    def greedy_inference(self, enc_output, vocabulary, max_length):
        """Reference transformer.py:141-175: list of B space-joined captions."""
        with torch.no_grad():
            ids, lengths = self.greedy_ids(enc_output, vocabulary.stoi["<SOS>"], vocabulary.stoi["<EOS>"], max_length)
        itos = vocabulary.itos
        return [" ".join(itos[i] for i in row) for row in engine.ids_to_lists(ids, lengths)]

    # -- beam search ---------------------------------------------------------------------------
    def beam_ids(self, enc_output, sos, eos, beam_width, max_length, return_scores=False, exact=False):
        """Best hypothesis per sample as token ids (ids int64 [B,max_length] padded with -1, lengths [B]) on the CPU.
        enc_output [B,S,E]: one memory per sample.  ``exact=True`` forces the full-prefix recompute for every sample
        (by default only samples that generated <PAD> take it)."""
        return engine.tf_beam_ids(self, enc_output, beam_width, max_length, sos, eos, return_scores=return_scores,
                                  exact=exact)

    def beam_search_inference(self, enc_output, batch_size, vocabulary, beam_width, max_length):
        """Reference transformer.py:177-244: list of ``batch_size`` space-joined captions.  ``enc_output`` is
        [batch_size * beam_width, S, E] with every sample's memory repeated beam_width times, as the reference's
        callers build it (cnnattn_model.py:65-66); [batch_size, S, E] is accepted too."""
        if enc_output.shape[0] == batch_size * beam_width and beam_width > 1:
            enc_output = enc_output[::beam_width]
        elif enc_output.shape[0] != batch_size:
            raise RuntimeError(f"capdec: enc_output has {enc_output.shape[0]} rows for batch_size {batch_size} x "
                               f"beam_width {beam_width}")
        with torch.no_grad():
            ids, lengths = self.beam_ids(enc_output, vocabulary.stoi["<SOS>"], vocabulary.stoi["<EOS>"], beam_width,
                                         max_length)
        itos = vocabulary.itos
        return [" ".join(itos[i] for i in row) for row in engine.ids_to_lists(ids, lengths)]
