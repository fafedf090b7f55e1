"""Checkpoint compatibility with the reference's ``.pth.tar`` files (utils.py:12-32, train.py:369-379).

A reference checkpoint is ``{"state_dict", "optimizer", "step"}``: the model's full state_dict (frozen backbone
included), ``torch.optim.Adam.state_dict()`` and the global step.  The classes of this package carry only the
trainable decoder + feature head, whose keys are a subset of the reference's, so loading drops the backbone keys
(exactly what ``utils.load_model`` does with its key filter) and saving can merge a backbone state back in.

Files are read with ``torch.load(..., weights_only=True)``: nothing in the file is executed.
"""
import torch

_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "positional_encoding.pe")


def _is_buffer(key):
    return key.endswith(_BUFFER_SUFFIXES)


def load_reference_checkpoint(path_or_dict, model, trainer=None):
    """Load a reference checkpoint into ``model`` (in place; with a ``CaptionTrainer`` the parameters are views of
    its flat buffer, which therefore receives the weights).  Returns the stored global step.

    With ``trainer`` the Adam moments are restored too.  The reference's optimizer state is keyed by parameter
    index in ``model.parameters()`` order of ITS model (backbone first); the index of a name is recovered as its rank
    among the state_dict keys that are not buffers (``running_*``, ``num_batches_tracked``, ``positional_encoding.pe``).
    Parameters without a stored state (the frozen backbone never gets one) are skipped."""
    ckpt = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu",
                                                                          weights_only=True)
    ref_sd = ckpt["state_dict"]
    core = getattr(model, "decoderRNN", None)
    own = model.state_dict()
    missing = [k for k in own if k not in ref_sd]
    if missing:
        raise RuntimeError(f"capdec: checkpoint lacks {len(missing)} keys of the model, e.g. {missing[:3]}")
    model.load_state_dict({k: ref_sd[k] for k in own})
    if trainer is not None and "optimizer" in ckpt:
        order = [k for k in ref_sd if not _is_buffer(k)]
        index_of = {k: i for i, k in enumerate(order)}
        state = ckpt["optimizer"].get("state", {})
        prefix = "decoderRNN." if core is not None else ""
        steps = set()
        for name, grad_view in trainer.grads.items():
            st = state.get(index_of.get(prefix + name, -1))
            if st is None:
                continue
            off = grad_view.data_ptr() - trainer.flat_g.data_ptr()
            a = off // 4
            n = grad_view.numel()
            trainer.m[a:a + n].copy_(st["exp_avg"].reshape(-1))
            trainer.v[a:a + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if steps:
            trainer.t = max(steps)
        groups = ckpt["optimizer"].get("param_groups")
        if groups:
            trainer.lr = float(groups[0].get("lr", trainer.lr))
    return int(ckpt.get("step", 0))


def save_reference_checkpoint(path, model, trainer=None, step=0, backbone_state=None):
    """Write ``{"state_dict", "optimizer", "step"}`` in the reference's schema.  ``backbone_state`` (the backbone
    entries of a reference state_dict, e.g. filtered from the checkpoint training started from) is merged in front
    so that the reference's strict ``load_state_dict`` accepts the file; without it the file holds the decoder and
    head only (loadable here, and by the reference after the same merge).  Returns the dict that was written."""
    sd = {}
    if backbone_state:
        sd.update({k: v.detach().cpu() for k, v in backbone_state.items()})
    sd.update({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    out = {"state_dict": sd, "step": int(step)}
    if trainer is not None:
        order = [k for k in sd if not _is_buffer(k)]
        prefix = "decoderRNN." if getattr(model, "decoderRNN", None) is not None else ""
        state = {}
        for name, grad_view in trainer.grads.items():
            a = (grad_view.data_ptr() - trainer.flat_g.data_ptr()) // 4
            n = grad_view.numel()
            state[order.index(prefix + name)] = {
                "step": torch.tensor(float(trainer.t)),
                "exp_avg": trainer.m[a:a + n].detach().cpu().reshape(grad_view.shape).clone(),
                "exp_avg_sq": trainer.v[a:a + n].detach().cpu().reshape(grad_view.shape).clone(),
            }
        out["optimizer"] = {
            "state": state,
            "param_groups": [{"lr": trainer.lr, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": 0,
                              "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                              "differentiable": False, "fused": None, "params": list(range(len(order)))}],
        }
    torch.save(out, path)
    return out
