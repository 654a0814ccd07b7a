"""The fine-tuning step around the hot path (the reference's script/run.py:40-90, BASELINE.json config 5).

  ranking_loss       run.py:66-77   self-adversarial binary cross entropy over (batch, 1 + num_negative) scores
  train_step         run.py:63-82   one step as the reference's loop runs it: forward in train() mode, loss, backward, optimiser
  GraphedTrainStep                  the same step -- forward, loss, backward, AdamW -- recorded ONCE into a hipGraph and replayed

Why a captured step: at FB15k237's size the step is ~ 300 short launches and the HOST sets its pace (issue time 6.0 ms of a
5.7 ms step, profiles/r5_finetune_phases.txt); a fine-tune is 340 k such steps on one static graph with one batch shape, so the
launch sequence never changes.  Per step the host then copies the (batch, 1 + num_negative, 3) triples into the graph's input and
replays.  The negatives are drawn outside the graph (tasks.prefetch_negatives: the sampler's own side stream, one batch ahead).

Multi-GPU (run.py:44-45 wraps the model in DistributedDataParallel): every rank captures forward + backward and the optimiser
step as TWO graphs and all-reduces ONE flat gradient bucket (675 KB for the ULTRA checkpoints) between them -- the same averaged
gradients DDP produces, without its per-bucket hooks inside a capture.
"""
import torch
from torch.nn import functional as F

from . import models, rspmm


def ranking_loss(pred, adversarial_temperature=1.0, num_negative=None):
    """run.py:66-77: column 0 is the positive; negatives are weighted by a softmax over their own scores (no gradient through
    the weights) or uniformly when the temperature is 0."""
    fused = _fused_loss(pred, adversarial_temperature, num_negative)
    if fused is not None:
        return fused
    target = torch.zeros_like(pred)
    target[:, 0] = 1
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction="none")
    neg_weight = torch.ones_like(pred)
    if adversarial_temperature > 0:
        with torch.no_grad():
            neg_weight[:, 1:] = F.softmax(pred[:, 1:] / adversarial_temperature, dim=-1)
    else:
        neg_weight[:, 1:] = 1 / (num_negative if num_negative is not None else pred.shape[1] - 1)
    loss = (loss * neg_weight).sum(dim=-1) / neg_weight.sum(dim=-1)
    return loss.mean()


FUSED_LOSS = True      # (A/B switch for tests: the torch op chain above is the other side)


class _RankingLoss(torch.autograd.Function):
    """The op chain of ranking_loss -- ~ 25 elementwise / reduction launches forward and backward on a (8, 257) tensor -- as one
    launch that returns the loss AND d loss / d pred (csrc/loss_kernels.hip)."""

    @staticmethod
    def forward(ctx, pred, temperature, uniform_weight):
        import ctypes
        from ._lib import check, lib
        pred = pred.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred)
        check(lib.ultra_ranking_loss(pred.data_ptr(), pred.shape[0], pred.shape[1], float(temperature), float(uniform_weight),
                                     loss.data_ptr(), grad.data_ptr(),
                                     ctypes.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)))
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        grad, = ctx.saved_tensors
        return grad * grad_out, None, None


def _fused_loss(pred, adversarial_temperature, num_negative):
    if not (FUSED_LOSS and pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 2 and 2 <= pred.shape[1] <= 4096
            and 1 <= pred.shape[0] <= 4096):
        return None
    uniform = 1.0 / (num_negative if num_negative is not None else pred.shape[1] - 1)
    return _RankingLoss.apply(pred, float(adversarial_temperature), uniform)


def make_adamw(model, lr=5e-4, capturable=False, **kwargs):
    """AdamW as config/transductive/inference.yaml:34-36 asks for it, in torch's single-launch implementation
    (fused=True); capturable=True keeps the step counters on the device so that the step can be recorded into a hipGraph."""
    return torch.optim.AdamW(model.parameters(), lr=lr, fused=True, capturable=bool(capturable), **kwargs)


def train_step(model, data, batch, optimizer, adversarial_temperature=1.0, num_negative=None):
    """One step of run.py:63-82 (the batch already carries its negatives: tasks.negative_sampling / prefetch_negatives).
    Returns the loss as a 0-d tensor (no host synchronisation)."""
    pred = model(data, batch)
    loss = ranking_loss(pred, adversarial_temperature, num_negative)
    loss.backward()
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach()


class GraphedTrainStep(object):
    """loss = GraphedTrainStep(model, data, optimizer, example_batch)(batch) for batches of example_batch's shape.

    `optimizer` must keep its state on the device (torch.optim.AdamW(..., fused=True, capturable=True): make_adamw(model,
    capturable=True)).  The warm-up runs that precede the capture are real steps on `example_batch`; parameters and optimiser
    state are put back to what they were afterwards, IN PLACE (the graph holds their addresses).

    process_group (or an initialised default group of more than one rank): the gradients are averaged over the ranks between
    the backward and the optimiser step -- one all-reduce of one flat bucket.

    What the graph points at stays alive with this object: the plans its warm-up asked for (pinned), the out-edge lists of the
    first layer's backward, the static input / loss tensors."""

    def __init__(self, model, data, optimizer, example_batch, adversarial_temperature=1.0, num_negative=None, warmup=3,
                 process_group=None):
        assert example_batch.is_cuda, "graph capture needs GPU tensors"
        self.model, self.data, self.optimizer = model, data, optimizer
        self.temperature, self.num_negative = adversarial_temperature, num_negative
        self.static_batch = example_batch.clone()
        self.group = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise ValueError("GraphedTrainStep: the optimizer must be built with capturable=True (train.make_adamw(model, "
                                 "capturable=True)): its step counters are read inside the captured graph")
        self._pinned = []
        self._capture(warmup)

    # ---- the pieces of a step ----
    def _forward_backward(self):
        pred = self.model(self.data, self.static_batch)
        loss = ranking_loss(pred, self.temperature, self.num_negative)
        loss.backward()
        return loss.detach()

    def _params(self):
        return [p for group in self.optimizer.param_groups for p in group["params"]]

    def _flatten_grads(self):
        grads = [p.grad.reshape(-1) for p in self._params() if p.grad is not None]
        if self._flat is None:
            self._flat = torch.cat(grads)
        else:
            torch.cat(grads, out=self._flat)

    def _unflatten_grads(self):
        grads = [p.grad for p in self._params() if p.grad is not None]
        pieces = self._flat.split([g.numel() for g in grads])
        torch._foreach_copy_(grads, [piece.view_as(g) for piece, g in zip(pieces, grads)])

    def _capture(self, warmup):
        model, opt = self.model, self.optimizer
        dev = self.static_batch.device
        model.train()
        params = self._params()
        saved_params = [p.detach().clone() for p in params]
        saved_state = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state.get(p, {}).items()}
                       for p in params}
        self._flat = None
        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), rspmm.record_plans() as used:
                for _ in range(max(int(warmup), 1)):
                    opt.zero_grad(set_to_none=True)
                    self._forward_backward()
                    if self.world > 1:
                        self._flatten_grads()
                        torch.distributed.all_reduce(self._flat, group=self.group)
                        self._flat.div_(self.world)
                        self._unflatten_grads()
                    opt.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._pinned = used.plans
            for plan in self._pinned:
                plan.pin(+1)
            # (the first layer's backward reads the graph's out-edge lists: an LRU cache owns them, so hold what it holds now)
            self._held = list(rspmm._OUT_CSR_CACHE.values())
            try:
                # the captured backward allocates the gradients from the graph's pool; the capture runs on the warm-up's stream
                # (autograd's AccumulateGrad nodes remember the stream they were made on)
                opt.zero_grad(set_to_none=True)
                models.CAPTURE_GENERIC_PATH = True
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                    self.static_loss = self._forward_backward()
                    if self.world > 1:
                        self._flatten_grads()
                    else:
                        opt.step()
                self.step_graph = None
                if self.world > 1:
                    self.step_graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.step_graph, pool=self.graph.pool(), stream=side, capture_error_mode="thread_local"):
                        self._flat.div_(self.world)
                        self._unflatten_grads()
                        opt.step()
            except BaseException:
                for plan in self._pinned:
                    plan.pin(-1)
                self._pinned = []
                raise
            finally:
                models.CAPTURE_GENERIC_PATH = False
            self.valid = getattr(model.entity_model, "_pending_valid", None) if hasattr(model, "entity_model") else None
            # parameters and optimiser state back to where they were before the warm-up steps, in place
            with torch.no_grad():
                for p, was in zip(params, saved_params):
                    p.copy_(was)
                for p in params:
                    state, was = opt.state.get(p, {}), saved_state[id(p)]
                    for k, v in state.items():
                        if torch.is_tensor(v):
                            if k in was:
                                v.copy_(was[k])
                            else:
                                v.zero_()      # (no state before the warm-up: moments and step counter start at zero)
            torch.cuda.synchronize()

    def __call__(self, batch):
        """One training step on `batch`; returns the loss (a 0-d tensor of the graph: read it before the next call)."""
        if batch.shape != self.static_batch.shape:
            raise ValueError("GraphedTrainStep was captured for batch shape %s, got %s"
                             % (tuple(self.static_batch.shape), tuple(batch.shape)))
        self.static_batch.copy_(batch, non_blocking=True)
        self.graph.replay()
        if self.step_graph is not None:
            torch.distributed.all_reduce(self._flat, group=self.group)
            self.step_graph.replay()
        return self.static_loss

    def check(self):
        """The assertion of models.py:196-197 for the LAST batch (a host synchronisation: call it when the loss is logged)."""
        if self.valid is not None:
            assert bool(self.valid.all()), "every row of `batch` must share its head (or tail) and its relation (models.py:196-197)"

    def __del__(self):
        try:
            for plan in self._pinned:
                plan.pin(-1)
        except Exception:
            pass
