"""Soft-attention LSTM caption decoder (Show-Attend-Tell style) over a grid or token memory.

NOT part of the reference repository: its "Attn" models are Transformer decoders.  The model is provided because
the project's goals name "a fused LSTM cell with additive attention over encoder tiles"; there is no reference
behaviour to match, so the formulas below ARE the definition (checked against a plain-PyTorch evaluation of the same
formulas in the tests, gradients against torch autograd on that evaluation).  Teacher-forced forward / backward and
greedy decoding.

    enc   = fc_feat(feats)                         [B,S,E]    per-cell / per-token projection
    pf    = att_f(enc)                             [B,S,A]    once per caption
    h0,c0 = init_h(mean_s enc), init_c(mean_s enc)
    step: u = att_h(h); e_s = att_v . tanh(u + pf_s); alpha = softmax_s(e); ctx = sum_s alpha_s enc_s
          h,c = LSTM([embed(token) | ctx], h, c);  logits = fc2(h)

All arithmetic runs on the gfx950 kernels of libcapdec.so: the projections on the MFMA contraction kernel, the
attention in one kernel per step (scores and weights in LDS, pf and enc read once), the fused LSTM cell, and the
vocabulary projection with fused arg-max.
"""
import ctypes

import torch
import torch.nn as nn

from capdec import _lib, engine, ops


class SoftAttentionLSTMDecoder(nn.Module):
    def __init__(self, feature_size, embed_size, hidden_size, attention_size, vocab_size):
        super().__init__()
        self.fc_feat = nn.Linear(feature_size, embed_size)
        self.att_f = nn.Linear(embed_size, attention_size)
        self.att_h = nn.Linear(hidden_size, attention_size)
        self.att_v = nn.Linear(attention_size, 1, bias=False)
        self.init_h = nn.Linear(embed_size, hidden_size)
        self.init_c = nn.Linear(embed_size, hidden_size)
        self.embed = nn.Embedding(vocab_size, embed_size)
        self.lstm = nn.LSTM(2 * embed_size, hidden_size, batch_first=True)
        self.fc2 = nn.Linear(hidden_size, vocab_size)

    # ---- shared preparation: memory rows, attention keys, initial state ----
    def _prepare(self, feats):
        if not feats.is_cuda or feats.dtype != torch.float32:
            raise RuntimeError("capdec: features must be an fp32 tensor on the GPU (HIP); there is no CPU path")
        from capdec import autograd as AG
        B, S, D = feats.shape
        enc, pf, _, h0, c0 = AG.satt_prepare(self, feats)
        return enc.view(B, S, -1), pf.view(B, S, -1), h0, c0

    def forward(self, feats, captions):
        """feats [B,S,D], captions [B,T] -> logits [B,T,V]; autograd-connected when gradients are enabled (forward and
        backward both on the HIP kernels, see capdec.autograd.satt_forward / satt_backward)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from capdec import autograd as AG
            return AG.satt_logits(self, feats, captions)
        enc, pf, h, c = self._prepare(feats)
        B, S, E = enc.shape
        T = captions.shape[1]
        A = pf.shape[2]
        lib = _lib.load()
        emb = ops.embed_pe(captions.t().contiguous(), self.embed.weight, None).view(T, B, E)
        xcat = torch.empty((B, 2 * E), device=feats.device, dtype=torch.float32)
        Hd = h.shape[1]
        hs = torch.empty((B, T, Hd), device=feats.device, dtype=torch.float32)
        w = (self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0, self.lstm.bias_hh_l0)
        v = self.att_v.weight.reshape(-1)
        for t in range(T):
            u = ops.linear(h, self.att_h.weight, self.att_h.bias)
            _lib.check(lib.capdec_additive_attention(u.data_ptr(), A, pf.data_ptr(), v.data_ptr(), enc.data_ptr(), B, S, A, E,
                                                     xcat.data_ptr() + 4 * E, 2 * E, None,
                                                     torch.cuda.current_stream().cuda_stream), "capdec_additive_attention")
            xcat[:, :E].copy_(emb[t])
            h, c = ops.lstm_step(xcat, h, c, *w, h_out=hs[:, t])
        return ops.linear(hs.view(B * T, Hd), self.fc2.weight, self.fc2.bias).view(B, T, -1)

    def caption_ids(self, feats, sos=1, eos=2, max_length=40):
        """Greedy decode: (ids int64 [B,max_length] padded with -1, lengths [B]) on the CPU."""
        self.eval()
        lib = _lib.load()
        with torch.no_grad():
            enc, pf, h0, c0 = self._prepare(feats)
            B, S, E = enc.shape
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            m = _lib.SattModel(E, h0.shape[1], pf.shape[2], self.embed.weight.shape[0], p(self.embed.weight),
                               p(self.lstm.weight_ih_l0), p(self.lstm.weight_hh_l0), p(self.lstm.bias_ih_l0),
                               p(self.lstm.bias_hh_l0), p(self.att_h.weight), p(self.att_h.bias), p(self.att_v.weight),
                               p(self.fc2.weight), p(self.fc2.bias))
            dev = feats.device
            out_ids = torch.empty((B, max_length), dtype=torch.int64, device=dev)
            out_len = torch.empty((B,), dtype=torch.int32, device=dev)
            if B == 0:
                return out_ids.cpu(), out_len.cpu()
            with torch.cuda.device(dev):
                nbytes = lib.capdec_satt_decode_workspace_bytes(ctypes.byref(m), B, max_length)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                _lib.check(lib.capdec_satt_greedy_decode(ctypes.byref(m), enc.data_ptr(), pf.data_ptr(), h0.data_ptr(),
                                                         c0.data_ptr(), B, S, max_length, sos, eos, ws.data_ptr(), nbytes,
                                                         out_ids.data_ptr(), out_len.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream),
                           "capdec_satt_greedy_decode")
            return out_ids.cpu(), out_len.cpu()

    def caption_images(self, feats, vocabulary, max_length=40):
        ids, lengths = self.caption_ids(feats, vocabulary.stoi["<SOS>"], vocabulary.stoi["<EOS>"], max_length)
        itos = vocabulary.itos
        return [" ".join(itos[i] for i in row) for row in engine.ids_to_lists(ids, lengths)]
