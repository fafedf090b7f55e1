"""Feature-projection heads shared by the Transformer wrappers: Linear -> dropout -> BatchNorm1d -> memory rows.

The reference repeats these few lines in every wrapper (cnnattn_model.py:27-30, vitattn_model.py:43-47,
vitcnnattn_model.py:47-53, yoloattn_model.py:70-74, yolocnnattn_model.py:75-81).  In eval mode the
BatchNorm is folded into the contraction's epilogue; in training mode it uses batch statistics.
"""
import torch

from capdec import functional as CF

BACKBONE_MSG = ("this package implements the caption-decoder hot path only: pass the frozen-backbone features with "
                "mode='precomputed' (the pretrained backbones need torchvision/ultralytics and network weights)")


def _features(images, mode):
    if mode != "precomputed":
        raise RuntimeError(BACKBONE_MSG)
    if not images.is_cuda:
        raise RuntimeError("capdec: features must live on the GPU (HIP); there is no CPU path")
    return images.to(torch.float32)


def single_head(model, fc, images, mode):
    """[B,D] -> [B,1,E]"""
    x = _features(images, mode)
    y = CF.linear_dropout_bn(model, [(fc, 0, x.shape[1])], x)
    return y.view(y.shape[0], 1, -1)


def pair_head(model, fc_a, split, images, mode):
    """[B,split+rest] -> [B,2,E]; memory row 0 is fc_a(x[:, :split]), row 1 is fc_cnn(x[:, split:])."""
    x = _features(images, mode)
    y = CF.linear_dropout_bn(model, [(fc_a, 0, split), (model.fc_cnn, split, x.shape[1])], x)
    return y.view(y.shape[0], 2, -1)
