"""YOLOCNN-Attn: [YOLO 86436 | Inception 2048] features -> two memory rows -> Transformer decoder.

Drop-in for the reference's ``models/yolocnnattn_model.py`` (yolocnnattn_model.py:51-102): same constructor, ``forward`` /
``caption_images`` signatures and ``state_dict`` keys (minus the frozen backbone, whose features are this
model's input).  All arithmetic runs on the gfx950 kernels of libcapdec.so.
"""
import torch
import torch.nn as nn

from models.modules import heads
from models.modules.transformer import DecoderTransformer


class YOLOCNNAttentionModel(DecoderTransformer):
    def __init__(self, embed_size, vocab_size, num_heads, num_layers, dropout=0.1, max_seq_length=50):
        super().__init__(embed_size, vocab_size, num_heads, num_layers, dropout, max_seq_length)
        self.yolo_output_size = 84 * 1029
        self.fc_yolo = nn.Linear(self.yolo_output_size, embed_size)
        self.fc_cnn = nn.Linear(2048, embed_size)
        self.dropout = nn.Dropout(dropout)
        self.batchnorm = nn.BatchNorm1d(2 * embed_size)

    def precompute_image(self, images):
        raise RuntimeError(heads.BACKBONE_MSG)

    def encode_features(self, images, mode="precomputed"):
        """Feature head: fc_yolo(x[:, :86436]) | fc_cnn(x[:, 86436:]) -> dropout -> BatchNorm1d(2E) -> [B,2,E]"""
        return heads.pair_head(self, self.fc_yolo, self.yolo_output_size, images, mode)

    def forward(self, images, captions, mode):
        return self.decoder_forward(self.encode_features(images, mode), captions)

    def caption_images(self, images, vocabulary, mode="precomputed", max_length=40):
        self.eval()
        with torch.no_grad():
            enc_output = self.encode_features(images, mode)
        return self.greedy_inference(enc_output, vocabulary, max_length)

    def caption_ids(self, images, sos=1, eos=2, mode="precomputed", max_length=40):
        """Greedy decode returning token ids: (ids int64 [B,max_length] padded with -1, lengths [B])."""
        self.eval()
        with torch.no_grad():
            enc_output = self.encode_features(images, mode)
            return self.greedy_ids(enc_output, sos, eos, max_length)

    def caption_images_beam_search(self, images, vocabulary, beam_width=3, mode="precomputed", max_length=50):
        self.eval()
        with torch.no_grad():
            enc_output = self.encode_features(images, mode)
        return self.beam_search_inference(enc_output, enc_output.shape[0], vocabulary, beam_width, max_length)

    def caption_ids_beam(self, images, sos=1, eos=2, beam_width=3, mode="precomputed", max_length=50):
        """Beam search returning token ids: (ids int64 [B,max_length] padded with -1, lengths [B])."""
        self.eval()
        with torch.no_grad():
            enc_output = self.encode_features(images, mode)
            return self.beam_ids(enc_output, sos, eos, beam_width, max_length)
