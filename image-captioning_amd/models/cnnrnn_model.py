"""CNN-RNN: pooled InceptionV3 features -> single-layer LSTM caption decoder.

Drop-in for the reference's ``models/cnnrnn_model.py`` (:6-110): same class names, constructor and
``forward`` / ``caption_images`` signatures, same ``state_dict`` keys under ``decoderRNN.`` (minus the
frozen backbone).  The ``torch.nn`` layers are parameter containers; all arithmetic runs on the gfx950
kernels of libcapdec.so (fused LSTM cell, vocabulary projection with fused arg-max).
"""
import torch
import torch.nn as nn

from capdec import engine
from capdec import functional as CF
from models.modules import heads


class DecoderRNN(nn.Module):
    def __init__(self, embed_size, hidden_size, vocab_size, dropout=0.2):
        super().__init__()
        self.hidden_size = hidden_size
        self.vocab_size = vocab_size
        self.fc1 = nn.Linear(2048, embed_size, bias=False)
        self.dropout = nn.Dropout(dropout)
        self.batchnorm = nn.BatchNorm1d(embed_size)
        self.embed = nn.Embedding(vocab_size, embed_size)
        self.lstm = nn.LSTM(embed_size, hidden_size, batch_first=True)
        self.fc2 = nn.Linear(hidden_size, vocab_size)

    def image_input(self, features):
        """dropout -> fc1 (no bias) -> BatchNorm1d: the LSTM's time-step-0 input [B,E] (reference :22-24)."""
        return CF.linear_dropout_bn(self, [(self.fc1, 0, features.shape[1])], features, dropout_first=True)

    def forward(self, features, captions):
        """features [B,2048], captions [B,T] -> logits [B,T,V] (reference :20-34)."""
        return CF.decoder_rnn_forward(self, features, captions)


class CNNtoRNN(nn.Module):
    def __init__(self, embed_size, hidden_size, vocab_size):
        super().__init__()
        self.decoderRNN = DecoderRNN(embed_size, hidden_size, vocab_size)

    def _features(self, images, mode):
        if mode != "precomputed":
            raise RuntimeError(heads.BACKBONE_MSG)
        if not images.is_cuda:
            raise RuntimeError("capdec: features must live on the GPU (HIP); there is no CPU path")
        return images.to(torch.float32)

    def forward(self, images, captions, mode='precomputed'):
        return self.decoderRNN(self._features(images, mode), captions)

    def precompute_image(self, images):
        raise RuntimeError(heads.BACKBONE_MSG)

    def caption_ids(self, images, sos=1, eos=2, mode="precomputed", max_length=40):
        """Greedy decode returning token ids: (ids int64 [B,max_length] padded with -1, lengths [B])."""
        self.eval()
        with torch.no_grad():
            x0 = self.decoderRNN.image_input(self._features(images, mode))
            return engine.rnn_greedy_ids(self.decoderRNN, x0, max_length, sos, eos)

    def caption_images(self, images, vocabulary, mode="precomputed", max_length=40):
        """Reference :57-110: list of B space-joined captions."""
        ids, lengths = self.caption_ids(images, vocabulary.stoi["<SOS>"], vocabulary.stoi["<EOS>"], mode, max_length)
        itos = vocabulary.itos
        return [" ".join(itos[i] for i in row) for row in engine.ids_to_lists(ids, lengths)]

    def caption_ids_beam(self, images, sos=1, eos=2, beam_width=3, mode="precomputed", max_length=50,
                         return_scores=False):
        """Beam search returning token ids: (ids int64 [B,max_length] padded with -1, lengths [B][, scores [B,W]])."""
        self.eval()
        with torch.no_grad():
            x0 = self.decoderRNN.image_input(self._features(images, mode))
            return engine.rnn_beam_ids(self.decoderRNN, x0, beam_width, max_length, sos, eos, return_scores)

    def caption_images_beam_search(self, images, vocabulary, beam_width=3, mode="precomputed", max_length=50):
        """Reference :112-209: list of B space-joined captions (best hypothesis, cut at its first <EOS>)."""
        ids, lengths = self.caption_ids_beam(images, vocabulary.stoi["<SOS>"], vocabulary.stoi["<EOS>"], beam_width,
                                             mode, max_length)
        itos = vocabulary.itos
        return [" ".join(itos[i] for i in row) for row in engine.ids_to_lists(ids, lengths)]
