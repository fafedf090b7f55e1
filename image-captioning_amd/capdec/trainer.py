"""Teacher-forced training step on the HIP kernels with one flat parameter / gradient buffer.

Reproduces the reference's inner loop (train.py:252-262: forward, CrossEntropyLoss(ignore_index=<PAD>),
backward, Adam) and its data-parallel hook (train.py:176,224-226,261: DDP averages the per-rank
gradients; every rank's loss is the mean over its OWN non-PAD targets and its BatchNorm uses its OWN
batch statistics).  Differences in mechanism, not in result:
  * parameters are views of one flat fp32 buffer and gradients of another, laid out in the order the
    backward pass completes them, so the gradient exchange is a handful of large RCCL all-reduces issued
    while the rest of the backward pass is still running, and Adam is one kernel over the flat buffers;
  * the per-step scalars (dropout seed, Adam's step count and bias corrections, learning rate) live in device
    memory (``ops.TrainState``) and are advanced by a kernel, so a step has no per-step kernel arguments and the whole
    of it -- forward, loss, backward, Adam -- can be captured once in a HIP graph and replayed (``graph=True``): one
    host call per step instead of ~150 kernel launches from Python, which is what bounds small per-GPU batches.
"""
import collections
import os

import torch
import torch.distributed as dist

from . import autograd as AG
from . import dist as cdist
from . import ops
from .autograd import Tape


def captions_strided_ok(ids, tgt):
    """Whether the two views can be read in place by the compaction and embedding kernels."""
    return ids.dim() == 2 and ids.stride(1) == 1 and tgt.stride(1) == 1 and ids.stride(0) == tgt.stride(0)


# batch rows (samples x positions) from which a launched step puts its weight gradients on the second stream
SIDE_STREAM_MIN_ROWS = int(os.environ.get("CAPDEC_WGRAD_STREAM_MIN_ROWS", "4096"))


def _second_stream(dev, others=()):
    """A stream whose kernels really run beside the current stream's (and beside those of ``others``).  HIP multiplexes
    its streams onto a few hardware queues (four by default), and streams that share one run their kernels one after
    the other: with an RCCL process group created first, the first stream torch's pool handed out sat on the queue of
    the default stream, and the 0.75 ms vocabulary weight gradient no longer overlapped the layers' backward (6.5
    instead of 5.6 ms per 512-sample step).  So candidates are tried: one spin kernel on every stream of the set takes
    the time of one when they all overlap, of two or more when two of them share a queue."""
    main = torch.cuda.current_stream(dev)
    cycles = 1_000_000           # 0.43 ms on an MI355X: one spin alone 0.09 ms at 200 k cycles, 0.11 beside another, 0.18 behind it

    def spin_ms(streams):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        a.record(main)
        torch.cuda._sleep(cycles)
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
        for st in streams:
            main.wait_stream(st)
        b.record(main)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b)

    with torch.cuda.device(dev):
        spin_ms(())
        alone = min(spin_ms(()), spin_ms(()))
        best = None
        for _ in range(12):
            cand = torch.cuda.Stream(device=dev)
            t = min(spin_ms(tuple(others) + (cand,)), spin_ms(tuple(others) + (cand,)))
            if t < 1.5 * alone:
                return cand
            if best is None or t < best[0]:
                best = (t, cand)
    return best[1]


def _head_spec(model):
    """(parts, names, dropout_first) of the model's feature head, mirroring models/modules/heads.py."""
    if hasattr(model, "fc_yolo") and hasattr(model, "fc_cnn"):
        return ("fc_yolo", model.yolo_output_size), "fc_cnn"
    if hasattr(model, "fc_vit") and hasattr(model, "fc_cnn"):
        return ("fc_vit", model.vit_out_size), "fc_cnn"
    for n in ("fc1", "fc_vit", "fc_yolo"):
        if hasattr(model, n):
            return (n, None), None
    raise RuntimeError("capdec: model has no feature head")


class CaptionTrainer:
    """One object per process (one process per GPU)."""

    # by-value parts of the dropout seeds: they tell the sites of one step apart (the per-step part is on the device)
    _SEED_HEAD, _SEED_DECODER = 0x2545F4914F6CDD1D, 0

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, pad_idx=0, overlap=True,
                 skip_dead_rows=True, wgrad_stream=None, graph=False, exercise_collectives=False, graph_cache=8):
        """overlap=True: one all-reduce per gradient segment on a communication stream of its own, issued as soon as the
        backward pass has enqueued that segment; overlap=False: a single all-reduce of the whole flat gradient buffer after the backward pass.
        skip_dead_rows: run the Transformer stack on the live prefix of every caption only (the positions from a
        caption's last counted target on feed nothing but ignored loss terms); same loss and gradients, the dropout
        masks are drawn per live row.  Applies where ``autograd.ragged_supported`` holds, else the full rows run.
        wgrad_stream: run the weight-gradient contractions on a second HIP stream beside the input-gradient chain
        (default on; the environment variable CAPDEC_WGRAD_STREAM=0 switches it off when the argument is None).
        graph: capture the step in a HIP graph.  The first step with a given input shape runs launch by launch (it
        also lets every kernel variant do its one-time set-up), the second is captured and replayed, later ones are
        replays (inputs are copied into fixed buffers first).  Same kernels in the same order as the launch-by-launch step, hence the
        same losses and parameters.  With more than one rank under RCCL the per-segment all-reduces are captured with the
        step (each is issued as soon as the backward pass has enqueued its segment, on RCCL's own stream, so it runs
        beside the rest of the backward pass when the graph replays) and Adam follows the last of them inside the graph;
        if the capture of a collective is refused, or under gloo (host-staged collectives), the graph holds forward +
        backward only and one all-reduce of the flat gradient buffer and Adam are launched after the replay.  At most
        ``graph_cache`` captured shapes are kept (the reference pads every batch to its own longest caption,
        get_loader.py:184, so the shape changes from batch to batch: pad or bucket the captions to few lengths, or the
        step mostly runs launch by launch); the least recently used one is dropped.
        exercise_collectives: issue every broadcast / all-reduce even in a one-rank process group (where each is the
        identity), so that the device-collective code path -- the communication stream, segment boundaries, stream
        ordering -- can run on a single GPU under RCCL."""
        self.model = model
        self.overlap = overlap
        self.skip_dead_rows = skip_dead_rows
        self._want_side = (os.environ.get("CAPDEC_WGRAD_STREAM", "1") == "1") if wgrad_stream is None else bool(wgrad_stream)
        self.side_stream = None
        self.betas, self.eps, self.pad_idx = betas, eps, pad_idx
        self.graph = bool(graph)
        self._graphs = collections.OrderedDict()   # input shapes -> None (seen once, run eagerly) | captured step
        self._graph_cache = max(1, int(graph_cache))
        self._capture_collectives = True           # until a capture with the all-reduces inside has been refused
        self.collectives_in_graph = False          # whether the last captured step holds its all-reduces
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # gloo has no dependable device collectives: stage through the host there (tests and rehearsals on one GPU);
        # RCCL ("nccl") reduces the device buffers in place
        self.coll = self.world > 1 or (bool(exercise_collectives) and dist.is_available() and dist.is_initialized())
        self.host_staged = self.coll and dist.get_backend(process_group) == "gloo"
        self._t, self._lr = 0, float(lr)
        self.rnn = getattr(model, "decoderRNN", None)          # CNN-RNN: the trainable module is the LSTM decoder
        core = self.rnn if self.rnn is not None else model
        self.core = core
        named = dict(core.named_parameters())
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("capdec: the trainer needs the model on the GPU (HIP); there is no CPU path")
        if self._want_side:
            self.side_stream = _second_stream(dev)
        # the gradient all-reduces of the overlapped step run on a stream of this trainer's choosing (see _reduce_segment)
        self.comm_stream = None
        if self.coll and not self.host_staged:
            self.comm_stream = _second_stream(dev, (self.side_stream,) if self.side_stream is not None else ())
        # the first step's seed comes from torch's CPU generator, so torch.manual_seed makes a run repeatable
        self.state = ops.TrainState(dev, self._lr, AG.draw_seed(), 0)
        # segments in the order the backward pass finishes them
        seg_names = [[n for n in named if n.startswith("fc2.")]]
        if self.rnn is not None:
            seg_names.append([n for n in named if n.startswith("lstm.")] + ["embed.weight"])
        else:
            (self.fc_a, self.split), self.fc_b = _head_spec(model)
            for l in range(len(model.decoder_layers) - 1, -1, -1):
                names = [n for n in named if n.startswith(f"decoder_layers.{l}.")]
                # the self-attention's three projection weights back to back, then their biases: the backward pass
                # treats them as one stacked layer (autograd._stacked_projections)
                first = [f"decoder_layers.{l}.self_attn.{w}.{k}" for k in ("weight", "bias") for w in ("W_q", "W_k", "W_v")]
                first = [n for n in first if n in named]
                seg_names.append(first + [n for n in names if n not in first])
            seg_names.append(["decoder_embedding.weight"])
        seg_names.append([n for n in named if n.split(".")[0] in ("fc1", "fc_vit", "fc_cnn", "fc_yolo", "batchnorm")])
        flat_names = [n for seg in seg_names for n in seg]
        missing = set(named) - set(flat_names)
        if missing:
            raise RuntimeError(f"capdec: parameters outside the decoder hot path: {sorted(missing)}")
        total = sum((named[n].numel() + 3) // 4 * 4 for n in flat_names)
        self.flat_p = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grads, self.segments = {}, []
        off = 0
        with torch.no_grad():
            for seg in seg_names:
                start = off
                for n in seg:
                    prm = named[n]
                    k = prm.numel()
                    view = self.flat_p[off:off + k].view_as(prm)
                    view.copy_(prm)
                    prm.data = view                       # the module now reads and updates the flat buffer
                    self.grads[n] = self.flat_g[off:off + k].view_as(prm)
                    prm.grad = self.grads[n]
                    off += (k + 3) // 4 * 4               # keep every tensor 16-byte aligned
                self.segments.append((start, off))
            self.flat_p[off:].zero_()
        # BatchNorm running statistics as views of one small buffer: DDP (broadcast_buffers=True, the default the
        # reference gets from accelerate) re-broadcasts rank 0's buffers at the start of every forward pass
        bn = core.batchnorm
        C = bn.running_mean.numel()
        self.flat_buf = torch.empty(2 * C, device=dev, dtype=torch.float32)
        with torch.no_grad():
            self.flat_buf[:C].copy_(bn.running_mean)
            self.flat_buf[C:].copy_(bn.running_var)
            bn.running_mean.data = self.flat_buf[:C]
            bn.running_var.data = self.flat_buf[C:]
        if self.coll:
            self._bcast(self.flat_p)                              # DDP starts from rank 0's weights
            self._bcast(self.flat_buf)

    # -----------------------------------------------------------------------------------------
    def _bcast(self, t):
        if self.host_staged:
            h = t.cpu()
            dist.broadcast(h, src=0, group=self.pg)
            t.copy_(h)
        else:
            dist.broadcast(t, src=0, group=self.pg)

    def _reduce_segment(self, i, works):
        if self._reduce_on:
            a, b = self.segments[i]
            side = self.side_stream
            try:
                if self.host_staged:
                    AG.join_side_stream()              # the segment's weight gradients may still be in flight there
                    h = self.flat_g[a:b].cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
                    self.flat_g[a:b].copy_(h)
                else:
                    # The segment's weight gradients are in flight on the side stream, its bias and LayerNorm gradients
                    # were written on this one.  The collective runs on a third stream that waits for both, so this
                    # stream - the input-gradient chain of the layers still to come - is not held up behind the weight
                    # gradients (joining the streams here instead cost 0.2 ms of a 5.6 ms step).  A collective called
                    # with async_op=False runs on the stream that is current at the call (seen in a kernel trace of
                    # this torch build: the copy of a one-rank all_gather lands on that stream, while async_op=True
                    # puts it on the process group's own stream - which shared a hardware queue with the default
                    # stream there, so the all-reduce would have run in line with the kernels it is meant to overlap).
                    # The third stream is picked like the second: its kernels run beside those of the other two.
                    # Before Adam this stream waits for it (_wait).
                    comm = self.comm_stream
                    if side is not None:
                        comm.wait_stream(side)
                    comm.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(comm):
                        dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.pg)
                    works.append(comm)
            except Exception as e:
                raise RuntimeError(f"capdec: gradient all-reduce failed on rank {dist.get_rank(self.pg)}, segment {i} "
                                   f"of {len(self.segments)} (elements {a}:{b}): {e}") from e

    def _wait(self, works):
        if works:
            # every segment's all-reduce is on the one communication stream, in order: Adam follows the last of them
            torch.cuda.current_stream().wait_stream(works[-1])

    # ---- per-step scalars: host shadows of the device state ---------------------------------------------
    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        self.state.set(lr=self._lr)

    @property
    def t(self):
        """Number of optimizer steps taken (Adam's step count)."""
        return self._t

    @t.setter
    def t(self, value):
        self._t = int(value)
        self.state.set(step=self._t)

    # ---- one step --------------------------------------------------------------------------------------------
    def _forward_backward(self, feats, captions, reduce_inside):
        """Everything of a step up to, not including, the optimizer: state advance, zeroed gradients, forward, loss,
        backward.  reduce_inside: issue the per-segment all-reduces while the backward pass is enqueued."""
        self.state.advance(self.betas[0], self.betas[1])
        self.flat_g.zero_()
        # inputs and targets as two views of the caption matrix (train.py:253-256); the Transformer path on live rows
        # reads them in place, every other path gets packed copies
        ids, tgt = captions[:, :-1], captions[:, 1:]
        self._reduce_on = reduce_inside
        # The second stream pays when a step's kernels are long enough for the host to stay ahead of both streams.  A
        # launched step of few rows is bound by the host's enqueueing, and the stream switches and events of the second
        # stream add to exactly that: 64 samples x 20 positions 2.52 ms with it against 1.64 without, 128 samples 2.55
        # against 2.30; from 256 samples on it wins (3.32 against 3.64 ms, 512 samples 5.45 against 5.87).  A captured
        # step keeps it at any size (replayed, 64 samples: 1.64 against 1.66 ms).
        rows = ids.shape[0] * ids.shape[1]
        side = self.side_stream if (rows >= SIDE_STREAM_MIN_ROWS or torch.cuda.is_current_stream_capturing()) else None
        AG.use_side_stream(side)
        try:
            if self.rnn is not None:
                loss = self._rnn_step(feats, ids.contiguous(), tgt.contiguous())
            else:
                loss = self._tf_step(feats, ids, tgt)
            AG.join_side_stream()
        finally:
            AG.use_side_stream(None)
        return loss

    def _all_reduce_whole(self):
        try:
            if self.host_staged:
                h = self.flat_g.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
                self.flat_g.copy_(h)
            else:
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)
        except Exception as e:
            raise RuntimeError(f"capdec: gradient all-reduce failed on rank {dist.get_rank(self.pg)} "
                               f"(whole buffer, {self.flat_g.numel() * 4} bytes): {e}") from e

    def _adam(self):
        # DDP semantics: mean of the per-rank gradients
        ops.adam_step_dev(self.flat_p, self.flat_g, self.m, self.v, self.state, self.betas[0], self.betas[1], self.eps,
                          grad_scale=1.0 / self.world)

    def step(self, feats, captions):
        """feats: precomputed backbone features [B,D]; captions [B,T+1] int64 (inputs [:, :-1], targets [:, 1:]).
        Returns the rank-local loss as a 1-element device tensor (no host sync)."""
        self.model.train()
        self._t += 1
        if self.coll:
            self._bcast(self.flat_buf)
        if self.graph:
            return self._graph_step(feats, captions)
        loss = self._forward_backward(feats, captions, reduce_inside=self.coll and self.overlap)
        if self.coll and not self.overlap:
            self._all_reduce_whole()
        self._adam()
        return loss

    def _capture(self, cap, inside):
        """Capture one step into cap.graph.  inside: with the per-segment all-reduces and Adam (device collectives)."""
        cap.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cap.graph):
            cap.loss = self._forward_backward(cap.feats, cap.captions, reduce_inside=inside)
            if inside or not self.coll:
                self._adam()
        cap.inside = inside

    def _graph_step(self, feats, captions):
        key = (tuple(feats.shape), tuple(captions.shape), feats.dtype, captions.dtype)
        if key not in self._graphs:
            # first sight of this shape: launch by launch (a real step; also every kernel variant's one-time set-up)
            self._graphs[key] = None
            while len(self._graphs) > self._graph_cache:
                self._graphs.popitem(last=False)           # least recently used shape: its graph and pool are freed
            loss = self._forward_backward(feats, captions, reduce_inside=self.coll and self.overlap)
            if self.coll and not self.overlap:
                self._all_reduce_whole()
            self._adam()
            return loss
        self._graphs.move_to_end(key)
        cap = self._graphs[key]
        if cap is None:
            cap = Tape()
            cap.feats = torch.empty_like(feats)
            cap.captions = torch.empty_like(captions)
            inside = self.coll and not self.host_staged and self.overlap and self._capture_collectives
            try:
                self._capture(cap, inside)
            except RuntimeError:
                if not inside:
                    raise
                # a collective that cannot be captured on this stack: keep the exchange outside the graph from now on
                self._capture_collectives = False
                torch.cuda.synchronize()
                self._capture(cap, False)
            self.collectives_in_graph = cap.inside
            self._graphs[key] = cap
        cap.feats.copy_(feats)
        cap.captions.copy_(captions)
        cap.graph.replay()
        loss = cap.loss.clone()            # the graph writes the same storage at every replay
        if self.coll and not cap.inside:
            self._all_reduce_whole()
            self._adam()
        return loss

    def _rnn_step(self, feats, ids, tgt):
        """CNN-RNN (reference cnnrnn_model.py:20-34): dropout -> fc1 -> BatchNorm, LSTM over [image, captions]."""
        rnn = self.rnn
        p = rnn.dropout.p
        sdev = self.state.seed_dev
        x0, htape = AG.head_forward(rnn, [(rnn.fc1, 0, feats.shape[1])], feats, True, self._SEED_HEAD, seed_dev=sdev)
        h, tape = AG.rnn_forward(rnn, x0, ids, p, self._SEED_DECODER, seed_dev=sdev)
        loss, saved = AG.vocab_loss(rnn, h, tgt.t().contiguous(), self.pad_idx)      # rows of h are time-major
        works = []
        dh = AG.vocab_loss_bwd(rnn, h, saved, self.grads)
        self._reduce_segment(0, works)
        dx0 = AG.rnn_backward(rnn, tape, dh, self.grads)
        self._reduce_segment(1, works)
        AG.head_backward(rnn, ["fc1"], htape, dx0, self.grads)
        self._reduce_segment(2, works)
        self._wait(works)
        return loss

    def _tf_step(self, feats, ids, tgt):
        model = self.model
        p = model.dropout.p
        # ---- forward ----
        parts = [(getattr(model, self.fc_a), 0, self.split if self.fc_b else feats.shape[1])]
        names = [self.fc_a]
        if self.fc_b:
            parts.append((getattr(model, self.fc_b), self.split, feats.shape[1]))
            names.append(self.fc_b)
        sdev = self.state.seed_dev
        y, htape = AG.head_forward(model, parts, feats, False, self._SEED_HEAD, seed_dev=sdev)
        B = feats.shape[0]
        enc = y.view(B, len(parts), -1)
        rg = None
        if self.skip_dead_rows and AG.ragged_supported(model, enc.shape[1], ids.shape[1], B) and captions_strided_ok(ids, tgt):
            rg = ops.RaggedRows(ids, tgt, self.pad_idx)
        else:
            ids, tgt = ids.contiguous(), tgt.contiguous()
        x, tape = AG.tf_forward(model, enc, ids, p, self._SEED_DECODER, ragged=rg, seed_dev=sdev)
        if rg is not None:
            loss, saved = AG.vocab_loss(model, x, rg.targets_c, self.pad_idx, rows=rg.rows)
        else:
            loss, saved = AG.vocab_loss(model, x, tgt, self.pad_idx)
        # ---- backward, gradient exchange overlapped segment by segment ----
        works = []
        dx = AG.vocab_loss_bwd(model, x, saved, self.grads)
        self._reduce_segment(0, works)
        denc = AG.tf_backward(model, tape, dx, self.grads, need_denc=True,
                              on_layer_done=(lambda l: self._reduce_segment(len(model.decoder_layers) - l, works))
                              if self._reduce_on else None)
        self._reduce_segment(len(self.segments) - 2, works)          # embedding
        AG.head_backward(model, names, htape, denc.reshape(B, -1), self.grads)
        self._reduce_segment(len(self.segments) - 1, works)          # head
        self._wait(works)
        return loss

    def scheduler_step(self, epoch, step_size=5, gamma=0.8):
        """``StepLR(step_size=5, gamma=0.8)`` stepped once per epoch (reference train.py:218,366): call after epoch
        number ``epoch`` (0-based) has finished; returns the learning rate of the next epoch."""
        if (epoch + 1) % step_size == 0:
            self.lr = self.lr * gamma          # the setter also writes the device-resident value the Adam kernel reads
        return self.lr

    def grad_bytes(self):
        return self.flat_g.numel() * 4
