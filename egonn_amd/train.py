"""Training-mode EgoNN graph on the MI355X operators (BASELINE configs[3]: the sharded training step).

The reference trains through MinkowskiEngine's autograd (training/trainer.py:160-175: `model.train()`,
`y = model(batch)`, `loss.backward()`, `optimizer.step()`).  Here `MinkGL.forward` in train mode builds the same
graph (models/minkgl.py:136-153 trunk, layers/eca_block.py:56-73 block, models/minkgl.py:46-60 head,
:207-225 decoder, layers/pooling.py:82-86 GeM) out of `torch.autograd.Function`s whose forward AND backward are
libegonn_hip kernels; PyTorch only owns the tensors, the tape and the optimiser.  Vectors of C or (B, C) values
(batch-norm statistics, the ECA gate's Conv1d over channels, GeM's exponent) are combined with tiny tensor ops.

BatchNorm uses the statistics of ALL rows of the batch (`nn.BatchNorm1d` on SparseTensor.F); with a process group
the per-channel sums are all-reduced (SyncBN) so that a sharded batch reproduces the single-GPU
statistics.  There is no CPU path: every Function needs the HIP library.

Both branches are differentiable: the global one (trunk + global head + decoder + GeM) for the batch-hard triplet
loss of models/loss.py, the local one (local head, descriptor decoder + L2 norm, keypoint regressor + tanh + the
quantiser's keypoint_position, sigma regressor + softplus) so that the reference's own local losses
(models/loss_utils.py, plain torch code on the output lists) can back-propagate into it.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SOFTPLUS = 0, 1, 2, 3


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- sparse convolution
class SparseConvFn(Function):
    """ME.MinkowskiConvolution / MinkowskiConvolutionTranspose on the cached maps of the current plan."""

    @staticmethod
    def forward(fctx, x, kernel, ctx: _lib.Context, level_in: int, level_out: int, ks: int, transposed: bool):
        k = _c(kernel.detach())
        if transposed:
            out = ctx.conv_transpose(level_in, x, k)
        else:
            out = ctx.conv(level_in, level_out, ks, x, k)
        fctx.save_for_backward(x, kernel)
        fctx.meta = (ctx, level_in, level_out, ks, transposed)
        return out

    @staticmethod
    def backward(fctx, g):
        x, kernel = fctx.saved_tensors
        ctx, lin, lout, ks, transposed = fctx.meta
        g = _c(g)
        k = kernel.detach()
        dx = dk = None
        if x is not None and fctx.needs_input_grad[0]:
            if ks == 1:                                   # dX = dY @ W^T, W (cin, cout) read as an (out=cin, in=cout) matrix
                dx = ctx.dense(g, _c(k), out_in=True)
            elif ks in (2, 3):
                # The input-gradient convolutions run on the fp16-split kernels like the forward ones, and gradients are SMALL
                # (1e-6 .. 1e-8 is ordinary): an fp16 part flushes below 2^-25 and carries 2^-25 absolute error below 2^-14.  The
                # library therefore scales this operand by a power of two per launch (max |g| -> [2^13, 2^14), exact, undone in
                # the epilogue: egonn_ctx_set_operand_autoscale) — the treatment the kernels give their weights.
                ctx.set_operand_autoscale(True)
                try:
                    if ks == 3:                           # nbr[o][k] = j  <=>  nbr[j][26-k] = o
                        dx = ctx.conv(lin, lin, 3, g, _c(k.flip(0).transpose(1, 2)))
                    elif not transposed:                  # strided conv  <->  transposed conv on the same map
                        dx = ctx.conv_transpose(lout, g, _c(k.transpose(1, 2)))
                    else:
                        dx = ctx.conv(lout, lin, 2, g, _c(k.transpose(1, 2)))
                finally:
                    ctx.set_operand_autoscale(False)
            else:
                raise NotImplementedError(f"input gradient of a k={ks} convolution")
        if fctx.needs_input_grad[1]:
            dk = ctx.conv_backward_weight(lin, lout, ks, transposed, x, g, tuple(kernel.shape))
        return dx, dk, None, None, None, None, None


def sparse_conv(ctx, x, conv_module, level_in, level_out):
    return SparseConvFn.apply(x, conv_module.kernel, ctx, level_in, level_out, conv_module.kernel_size,
                              bool(conv_module.transpose))


# ----------------------------------------------------------------------------- batch norm (batch statistics)
def _all_reduce(t: torch.Tensor, group):
    if group is not None:
        from .distributed import all_reduce_sum
        all_reduce_sum(t, group)
    return t


def combine_batch_stats(sum_x: torch.Tensor, count: torch.Tensor, group=None):
    """(sum, count) of this rank -> global (mean, total count); one all-reduce of C+1 values."""
    buf = torch.cat([sum_x.double(), count.double().reshape(1)])
    _all_reduce(buf, group)
    total = buf[-1]
    return (buf[:-1] / total).float(), total


def level_totals(ctx: _lib.Context, group=None) -> List[float]:
    """rows of every level summed over the ranks of `group` (the N of the whole-batch BatchNorm statistics):
    one all-reduce of 8 values per step."""
    counts = [float(ctx.level_count(l)) for l in range(8)]
    if group is None:
        return counts
    t = torch.tensor(counts, dtype=torch.float64, device=ctx.device)
    _all_reduce(t, group)
    return t.tolist()


class BatchNormFn(Function):
    """nn.BatchNorm1d over all rows in train mode (MinkowskiBatchNorm), optional fused ReLU, optional SyncBN.
    Statistics: ONE pass of shifted sums (sum d, sum d^2 with d = x - running_mean: additive over ranks, robust
    against cancellation because the shift is close to the mean), all-reduced when `group` is given; the per-channel
    bookkeeping (mean, invstd, folded scale/shift, running statistics; A/B/C, dgamma, dbeta in backward) runs in one
    small kernel each."""

    @staticmethod
    def forward(fctx, x, weight, bias, ctx: _lib.Context, bn: torch.nn.BatchNorm1d, relu: bool, group, total):
        n, c = x.shape
        lib = ctx.lib
        track = bn.track_running_stats and bn.running_mean is not None
        shift_pt = bn.running_mean if track else torch.zeros(c, dtype=torch.float32, device=x.device)
        s = ctx.col_stats(3, x, mean=shift_pt)
        _all_reduce(s, group)
        total = float(total if total is not None else n)
        out4 = torch.empty((4, c), dtype=torch.float32, device=x.device)
        if track:
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
        else:
            mom = 0.0
        ctx._call(lib.egonn_bn_train_finalize, s.data_ptr(), shift_pt.data_ptr(), total, c, weight.data_ptr(),
                  bias.data_ptr(), float(bn.eps), float(mom), _lib._ptr(bn.running_mean if track else None),
                  _lib._ptr(bn.running_var if track else None), out4.data_ptr())
        if track:
            bn.num_batches_tracked += 1
        y = ctx.affine_act(x, out4[2], out4[3], relu)
        fctx.save_for_backward(x, y if relu else None, out4, weight)
        fctx.meta = (ctx, group, total)
        return y

    @staticmethod
    def backward(fctx, g):
        x, y, out4, weight = fctx.saved_tensors
        ctx, group, total = fctx.meta
        c = x.shape[1]
        g = _c(g)
        s = ctx.col_stats(2, g, b=x, mask=y, mean=out4[0])          # sum g', sum g' (x - mean)   (this rank's rows)
        sg = s
        if group is not None:
            sg = _all_reduce(s.clone(), group)                      # whole-batch sums for the input gradient
        out5 = torch.empty((5, c), dtype=torch.float32, device=x.device)
        ctx._call(ctx.lib.egonn_bn_backward_finalize, s.data_ptr(), sg.data_ptr(), total, c, weight.data_ptr(),
                  out4[0].data_ptr(), out4[1].data_ptr(), out5.data_ptr())
        dx = ctx.affine3(g, y, x, out5[0], out5[1], out5[2])
        return dx, out5[3], out5[4], None, None, None, None, None


def batch_norm(ctx, x, bn_module, relu: bool, group=None, total=None):
    bn = bn_module.bn
    return BatchNormFn.apply(x, bn.weight, bn.bias, ctx, bn, relu, group, total)


# ----------------------------------------------------------------------------- per-sample pooling / ECA tail
class SegmentMeanFn(Function):
    """ME.MinkowskiGlobalPooling (per-sample mean of the rows) -> (B, C)."""

    @staticmethod
    def forward(fctx, x, ctx: _lib.Context, level: int):
        fctx.meta = (ctx, level)
        return ctx.global_avg_pool(level, x)

    @staticmethod
    def backward(fctx, g):
        ctx, level = fctx.meta
        return ctx.segment_broadcast(level, _c(g), mean=True), None, None


class GateResidualFn(Function):
    """relu(x * gate[sample] + residual): MinkowskiBroadcastMultiplication + `out += residual` + MinkowskiReLU."""

    @staticmethod
    def forward(fctx, x, gate, residual, ctx: _lib.Context, level: int):
        """gate None: plain ME BasicBlock tail relu(x + residual)."""
        out = ctx.gate_residual(level, x, None if gate is None else _c(gate.detach()), residual, relu=True)
        fctx.save_for_backward(x, gate, out)
        fctx.meta = (ctx, level)
        return out

    @staticmethod
    def backward(fctx, g):
        x, gate, out = fctx.saved_tensors
        ctx, level = fctx.meta
        g = _c(g)
        dx, dres = ctx.gate_residual_backward(level, g, out, None if gate is None else _c(gate.detach()),
                                              want_residual=fctx.needs_input_grad[2])
        dgate = ctx.segment_sums(level, 2, g, b=out, x2=x) if gate is not None and fctx.needs_input_grad[1] else None
        return dx, dgate, dres, None, None


class EcaGateFn(Function):
    """sigmoid(Conv1d_k(mean)) over the channel axis of the (B, C) per-sample means (layers/eca_block.py:28-31)."""

    @staticmethod
    def forward(fctx, mean, weight, ctx: _lib.Context):
        w = _c(weight.detach().reshape(-1))
        B, c = mean.shape
        gate = torch.empty_like(mean)
        ctx._call(ctx.lib.egonn_eca_gate, mean.data_ptr(), w.data_ptr(), w.numel(), B, c, gate.data_ptr())
        fctx.save_for_backward(mean, weight, gate)
        fctx.meta = ctx
        return gate

    @staticmethod
    def backward(fctx, g):
        mean, weight, gate = fctx.saved_tensors
        ctx = fctx.meta
        w = _c(weight.detach().reshape(-1))
        B, c = mean.shape
        g = _c(g)
        dmean = torch.empty_like(mean)
        dw = torch.empty(w.numel(), dtype=torch.float32, device=mean.device)
        ctx._call(ctx.lib.egonn_eca_gate_backward, g.data_ptr(), gate.data_ptr(), mean.data_ptr(), w.data_ptr(), w.numel(),
                  B, c, dmean.data_ptr(), dw.data_ptr())
        return dmean, dw.reshape(weight.shape), None


def eca_tail(ctx, level, x, residual, eca_module):
    """layers/eca_block.py:21-36,66-73: gate = sigmoid(Conv1d_k(mean_b(x))), out = relu(x * gate + residual)."""
    m = SegmentMeanFn.apply(x, ctx, level)                                       # (B, C)
    gate = EcaGateFn.apply(m, eca_module.conv.weight, ctx)                       # fixed-order sums: deterministic
    return GateResidualFn.apply(x, gate, residual, ctx, level)


class AddFn(Function):
    @staticmethod
    def forward(fctx, a, b, ctx: _lib.Context):
        return ctx.add(a, b)

    @staticmethod
    def backward(fctx, g):
        return g, g, None


# ----------------------------------------------------------------------------- dense layers / GeM
class LinearFn(Function):
    """ME.MinkowskiLinear (+ fused MinkowskiReLU / Tanh / Softplus): act(rows @ W^T + b), W (out, in).
    `act`: True/False (ReLU or none) or an ACT_* code."""

    @staticmethod
    def forward(fctx, x, weight, bias, ctx: _lib.Context, act):
        act = int(act)
        y = ctx.dense(x, _c(weight.detach()), out_in=True, bias=None if bias is None else _c(bias.detach()), act=act)
        fctx.save_for_backward(x, weight, y if act else None)
        fctx.meta = (ctx, act, bias is not None)
        return y

    @staticmethod
    def backward(fctx, g):
        x, weight, y = fctx.saved_tensors
        ctx, act, has_bias = fctx.meta
        g = _c(g)
        if act:
            g = ctx.act_backward(act, g, y)
        dx = ctx.dense(g, _c(weight.detach()), out_in=False) if fctx.needs_input_grad[0] else None
        dw = ctx.dense_backward_weight(g, x) if fctx.needs_input_grad[1] else None
        db = ctx.col_stats(0, g)[0].clone() if has_bias and fctx.needs_input_grad[2] else None
        return dx, dw, db, None, None


class L2NormalizeFn(Function):
    """ME.MinkowskiFunctional.normalize (F.normalize over the channels of every row)."""

    @staticmethod
    def forward(fctx, x, ctx: _lib.Context):
        fctx.save_for_backward(x)
        fctx.meta = ctx
        return ctx.l2_normalize(x)

    @staticmethod
    def backward(fctx, g):
        (x,) = fctx.saved_tensors
        return fctx.meta.l2_normalize(x, _c(g)), None


class GeMFn(Function):
    """layers/pooling.py:82-86: (mean_b clamp(x, 1e-6)^p)^(1/p)."""

    @staticmethod
    def forward(fctx, x, p, ctx: _lib.Context, level: int):
        out = ctx.gem(level, x, p)
        fctx.save_for_backward(x, p, out)
        fctx.meta = (ctx, level)
        return out

    @staticmethod
    def backward(fctx, g):
        x, p, out = fctx.saved_tensors
        ctx, level = fctx.meta
        off = ctx.level_batch_offsets(level)
        cnt = torch.tensor([off[b + 1] - off[b] for b in range(ctx.batch_size)], dtype=torch.float32,
                           device=x.device).clamp_(min=1).unsqueeze(1)
        pv = p.detach().reshape(()).float()
        g = _c(g)
        dx = dp = None
        if fctx.needs_input_grad[0]:
            # an empty sample has out = 0: its coefficient is 0, not 0 * inf
            coef = torch.where(out > 0, g * out.clamp_min(1e-30).pow(1.0 - pv) / cnt, torch.zeros_like(out))
            dx = ctx.gem_backward(level, x, _c(coef), _c(p.detach().reshape(-1).float()))
        if fctx.needs_input_grad[1]:
            T = ctx.segment_sums(level, 1, x, p=_c(p.detach().reshape(-1).float()))    # sum_r t^p ln t
            S = out.pow(pv) * cnt                                                       # sum_r t^p
            dout_dp = out * (-(torch.log(out.pow(pv))) / (pv * pv) + T / (pv * S))
            dp = (g * dout_dp).sum().reshape(p.shape)
        return dx, dp, None, None


# ----------------------------------------------------------------------------- the graph
def trunk_forward(model, ctx, group=None) -> Dict[int, torch.Tensor]:
    """MinkTrunk.forward (reference models/minkgl.py:136-153) with all-ones input features."""
    t = model.trunk
    tot = level_totals(ctx, group)
    x = SparseConvFn.apply(None, t.convs['0'].kernel, ctx, 0, 0, t.convs['0'].kernel_size, False)
    x = batch_norm(ctx, x, t.bn['0'], True, group, tot[0])
    levels = {}
    for i in range(1, len(t.planes) + 1):
        x = sparse_conv(ctx, x, t.convs[str(i)], i - 1, i)
        x = batch_norm(ctx, x, t.bn[str(i)], True, group, tot[i])
        for blk in t.blocks[str(i)]:
            y = sparse_conv(ctx, x, blk.conv1, i, i)
            y = batch_norm(ctx, y, blk.norm1, True, group, tot[i])
            y = sparse_conv(ctx, y, blk.conv2, i, i)
            y = batch_norm(ctx, y, blk.norm2, False, group, tot[i])
            res = x
            if blk.downsample is not None:
                res = sparse_conv(ctx, x, blk.downsample[0], i, i)
                res = batch_norm(ctx, res, blk.downsample[1], False, group, tot[i])
            x = eca_tail(ctx, i, y, res, blk.eca)
        levels[i] = x
    return levels


def head_forward(head, ctx, levels: Dict[int, torch.Tensor]):
    """MinkHead.forward (reference models/minkgl.py:46-60)."""
    y = sparse_conv(ctx, levels[head.max_level], head.conv1x1[str(head.max_level)], head.max_level, head.max_level)
    for level in range(head.max_level - 1, head.min_level - 1, -1):
        y = sparse_conv(ctx, y, head.tconv[str(level + 1)], level + 1, level)
        if level in head.in_levels:
            y = AddFn.apply(y, sparse_conv(ctx, levels[level], head.conv1x1[str(level)], level, level), ctx)
    return head.min_level, y


def global_branch(model, ctx, group=None, levels=None) -> torch.Tensor:
    """trunk -> global head -> descriptor decoder -> GeM  (reference models/minkgl.py:269-287)."""
    if levels is None:
        levels = trunk_forward(model, ctx, group)
    lvl, x = head_forward(model.global_head, ctx, levels)
    net = model.global_descriptor_decoder.net
    x = LinearFn.apply(x, net[0].linear.weight, net[0].linear.bias, ctx, True)
    x = LinearFn.apply(x, net[2].linear.weight, net[2].linear.bias, ctx, False)
    method = getattr(model, "global_pool_method", "GeM")
    if method != "GeM":          # MAC / SPoC (layers/pooling.py:46-69) exist for eval-mode forwards only
        raise NotImplementedError(f"train-mode forward is implemented for GeM pooling; this model pools with {method!r} "
                                  "(use model.eval(), or train with pool_method='GeM' as config_egonn.txt does)")
    return GeMFn.apply(x, model.global_pooling.pooling.p, ctx, lvl)


def minkfpn_forward(fpn, ctx, group=None):
    """MinkFPN.forward in train mode (reference models/minkfpn.py:65-93; BasicBlock / ECABasicBlock, all-ones input
    features): the same graph as egonn_amd.minkloc.MinkFPN.run, on the differentiable operators."""
    tot = level_totals(ctx, group)

    def block(level, x, b):
        y = sparse_conv(ctx, x, b.conv1, level, level)
        y = batch_norm(ctx, y, b.norm1, True, group, tot[level])
        y = sparse_conv(ctx, y, b.conv2, level, level)
        y = batch_norm(ctx, y, b.norm2, False, group, tot[level])
        res = x
        if b.downsample is not None:
            res = sparse_conv(ctx, x, b.downsample[0], level, level)
            res = batch_norm(ctx, res, b.downsample[1], False, group, tot[level])
        if hasattr(b, 'eca'):
            return eca_tail(ctx, level, y, res, b.eca)
        return GateResidualFn.apply(y, None, res, ctx, level)

    assert fpn.conv0.kernel_size == 5 and fpn.conv0.kernel.shape[1] == 1, "train mode: k=5, 1-channel input layer"
    x = SparseConvFn.apply(None, fpn.conv0.kernel, ctx, 0, 0, 5, False)
    x = batch_norm(ctx, x, fpn.bn0, True, group, tot[0])
    fmaps = []
    if fpn.num_top_down == fpn.num_bottom_up:
        fmaps.append((0, x))
    level = 0
    for ndx, (conv, bn, blocks) in enumerate(zip(fpn.convs, fpn.bn, fpn.blocks)):
        x = sparse_conv(ctx, x, conv, level, level + 1)
        level += 1
        x = batch_norm(ctx, x, bn, True, group, tot[level])
        for b in blocks:
            x = block(level, x, b)
        if fpn.num_bottom_up - 1 - fpn.num_top_down <= ndx < len(fpn.convs) - 1:
            fmaps.append((level, x))
    x = sparse_conv(ctx, x, fpn.conv1x1[0], level, level)
    for ndx, tconv in enumerate(fpn.tconvs):
        x = sparse_conv(ctx, x, tconv, level, level - 1)
        level -= 1
        flevel, f = fmaps[-ndx - 1]
        assert flevel == level
        x = AddFn.apply(x, sparse_conv(ctx, f, fpn.conv1x1[ndx + 1], level, level), ctx)
    return level, x


def local_branch(model, ctx, levels: Dict[int, torch.Tensor]):
    """local head -> descriptor decoder (+L2 norm), keypoint regressor (+tanh), sigma regressor (+softplus)
    (reference models/minkgl.py:289-308).  Returns (level, descriptors, keypoint offsets, sigma), rows of `level`."""
    lvl, x = head_forward(model.local_head, ctx, levels)
    d = model.local_descriptor_decoder.net
    desc = LinearFn.apply(x, d[0].linear.weight, d[0].linear.bias, ctx, ACT_RELU)
    desc = LinearFn.apply(desc, d[2].linear.weight, d[2].linear.bias, ctx, ACT_NONE)
    if model.local_descriptor_decoder.normalize:
        desc = L2NormalizeFn.apply(desc, ctx)
    k = model.local_keypoint_regressor.net
    kp = LinearFn.apply(x, k[0].linear.weight, k[0].linear.bias, ctx, ACT_RELU)
    kp = LinearFn.apply(kp, k[2].linear.weight, k[2].linear.bias, ctx, ACT_TANH)
    g = model.local_sigma_regressor.net
    sg = LinearFn.apply(x, g[0].linear.weight, g[0].linear.bias, ctx, ACT_RELU)
    sg = LinearFn.apply(sg, g[2].linear.weight, g[2].linear.bias, ctx, ACT_SOFTPLUS)
    return lvl, desc, kp, sg


# ----------------------------------------------------------------------------- sharded step plumbing
def all_reduce_gradients(params: List[torch.nn.Parameter], group=None, world_size: Optional[int] = None):
    """Sum the gradients of all ranks in ONE flat RCCL all-reduce (4.71 M values = 18.8 MB for EgoNN; with the loss
    evaluated on the all-gathered embeddings each rank holds the contribution of its own scans, so the sum is the
    gradient of the whole batch — no averaging)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    from .distributed import all_reduce_sum
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    all_reduce_sum(flat, group)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


class TrainStep:
    """One optimisation step of the reference's global phase (training/trainer.py:157-175,193), sharded over the
    ranks of the default process group (BASELINE configs[3]: batch 256 = 32 scans per GPU on 8 GPUs):

        forward of this rank's scans (SyncBN statistics over the whole batch: one all-reduce of C+1 and one of C
        values per BatchNorm layer and direction)
        -> RCCL all-gather of the (b_local, 256) global descriptors (differentiable, distributed.py)
        -> batch-hard triplet loss on the gathered (B, 256) matrix with the (B, B) masks every rank holds
        -> backward (each rank back-propagates the rows it produced)
        -> ONE flat SUM all-reduce of the parameter gradients -> optimizer.step()

    With a single process it is exactly the reference's step."""

    def __init__(self, model, optimizer, margin: float = 0.2):
        from .loss import BatchHardTripletLossWithMasks
        self.model, self.optimizer = model, optimizer
        self.loss_fn = BatchHardTripletLossWithMasks(margin)

    def __call__(self, batch: Dict[str, torch.Tensor], positives_mask: torch.Tensor, negatives_mask: torch.Tensor,
                 step_optimizer: bool = True, shard_sizes=None):
        """shard_sizes: scans per rank when every rank knows them from the sampler (e.g. `distributed.shard_bounds` of the
        B = positives_mask.shape[0] scans): the embedding exchange is then ONE all-gather with no host synchronisation.
        Default (None): the sizes are exchanged first, so ANY sharding works and a mismatch can never leave some ranks
        inside a collective the others did not enter."""
        import torch.distributed as dist
        from .distributed import all_gather_embeddings
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        model = self.model
        model.train()
        model.sync_bn_group = dist.group.WORLD if sharded else None
        self.optimizer.zero_grad(set_to_none=True)
        y = model(batch, disable_local_head=True)
        emb = all_gather_embeddings(y['global'], shard_sizes if sharded else None)
        dev = emb.device
        loss, stats, _ = self.loss_fn(emb, positives_mask.to(dev), negatives_mask.to(dev))
        loss.backward()
        if sharded:
            all_reduce_gradients(list(model.parameters()))
        if step_optimizer:
            self.optimizer.step()
        return loss.detach(), stats
