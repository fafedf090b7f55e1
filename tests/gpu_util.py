"""Helpers shared by the GPU parity tests (HIP path vs the CPU oracle)."""
import numpy as np
import torch

from oracle import decoder_ref as R

REL = 1e-4   # north-star tolerance: fp32 logits / gradients within 1e-4 relative


def close(got, want, rtol=REL, atol=0.0, what=""):
    got = torch.as_tensor(np.asarray(got.detach().cpu() if isinstance(got, torch.Tensor) else got)).double()
    want = torch.as_tensor(np.asarray(want.detach().cpu() if isinstance(want, torch.Tensor) else want)).double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(want.abs().max()) if want.numel() else 1.0
    err = float((got - want).abs().max()) if want.numel() else 0.0
    assert err <= atol + rtol * max(scale, 1e-30), f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / max(scale, 1e-30):.2e})"


def plain_decoder(E, V, H, L, max_len=50, dropout=0.0):
    """Concrete DecoderTransformer without a feature head."""
    from models.modules.transformer import DecoderTransformer

    class PlainDecoder(DecoderTransformer):
        def precompute_image(self, images):
            return images

        def forward(self, enc, captions):
            return self.decoder_forward(enc, captions)

        def caption_images(self, *a, **k):
            raise NotImplementedError

        def caption_images_beam_search(self, *a, **k):
            raise NotImplementedError

    return PlainDecoder(E, V, H, L, dropout, max_len)


def decoder_from_sd(sd, H, dropout=0.0):
    E = sd["decoder_embedding.weight"].shape[1]
    V = sd["decoder_embedding.weight"].shape[0]
    L = R.num_layers_of(sd)
    max_len = sd["positional_encoding.pe"].shape[1]
    dec = plain_decoder(E, V, H, L, max_len, dropout)
    missing, unexpected = dec.load_state_dict(sd, strict=False)
    assert not missing, missing
    return dec.cuda().eval()


def cpu_sd(module):
    return {k: v.detach().cpu() for k, v in module.state_dict().items()}


def check_greedy(got_rows, sd, enc_cpu, H, max_length, want_rows=None, tol=REL):
    """Token ids must be identical wherever the oracle's top-2 logit margin exceeds the tolerance.

    got_rows: list of id lists from the HIP path.  The oracle (full-prefix recompute) is run on the same
    inputs; on the first differing token of a row the oracle's logits for that prefix are recomputed and
    the gap between its best logit and the logit of the HIP path's token must be within tol * max|logit|
    (a near-tie); the rest of that row is then not comparable and is skipped.
    """
    if want_rows is None:
        want_rows = R.greedy_inference_ids(sd, enc_cpu, H, max_length)
    n_ties = 0
    for b, (g, w) in enumerate(zip(got_rows, want_rows)):
        if g == w:
            continue
        n = min(len(g), len(w))
        s = next((i for i in range(n) if g[i] != w[i]), n)
        prefix = torch.tensor([[R.SOS] + list(w[:s])], dtype=torch.long)
        logits = R.decoder_forward(sd, enc_cpu[b:b + 1], prefix, H, take_last=True)[0]
        tok_g = g[s] if s < len(g) else R.EOS
        gap = float(logits.max() - logits[tok_g])
        assert gap <= tol * float(logits.abs().max()), \
            f"row {b} step {s}: HIP token {tok_g} vs oracle {w[s] if s < len(w) else R.EOS}, logit gap {gap:.3e}"
        n_ties += 1
    return n_ties
