"""Data-parallel replicas: one process per GPU, gradients summed with ONE all-reduce per parameter
group (E+G after the E+G backward, D after each of the two D backwards) over RCCL/xGMI
(``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).  The reference has no
distributed code at all; semantics chosen here:

* every replica holds identical weights / Adam state and draws its own batch, z and Dropout2d masks;
* gradients are averaged (sum all-reduce, 1/world folded into the Adam kernel on the GPU path);
* BatchNorm uses local batch statistics; the running buffers are averaged once per iteration so that
  ``state_dict()`` stays identical on all ranks.
"""
import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks of one flat fp32 buffer (one collective per parameter group).  Issued whenever a
    process group exists -- also on a 1-rank group, where it is the identity but still exercises the backend."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def allreduce_sum_async_(flat: torch.Tensor, group=None):
    """Start the in-place sum of one flat buffer and return the work handle (``.wait()`` orders the caller's stream --
    or, with gloo, the host -- after it); whatever is launched before the wait overlaps with the collective."""
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)


def average_buffers_(buffers, group=None):
    """Average a list of small tensors (BatchNorm running stats) with a single collective."""
    if not (dist.is_available() and dist.is_initialized()) or not buffers:
        return
    w = world_size(group)
    flat = torch.cat([b.reshape(-1).float() for b in buffers])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / w)
    off = 0
    for b in buffers:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()


class GradSync:
    """Hook for ``image_scms.training_utils.ali_step(..., grad_sync=...)`` (autograd path, any device):
    flatten the gradients of a parameter list, all-reduce once, write the average back."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, params):
        w = world_size(self.group)
        if w <= 1:
            return
        params = [p for p in params if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        allreduce_sum_(flat, self.group)
        flat.mul_(1.0 / w)
        off = 0
        for p in params:
            p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
            off += p.numel()
