"""Data-parallel helpers: one process per GPU, torch.distributed ('nccl' == RCCL over xGMI on ROCm;
'gloo' in the CPU tests).  The hot path shards by image, so inference needs no data-path
collective; the only exchanges are (a) the gradient exchange of the Dense layer in training - the SUM all-reduce of
[dWt | db | loss] or, for equal shards, the all-gather of features and dz it can be computed from (engine.Trainer) - and
(b) an optional all-gather of fixed-size detection records.  These helpers are
device-agnostic on purpose: the same code runs under gloo with CPU tensors in tests/."""
import torch
import torch.distributed as dist


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n, rank, world_size):
    """Contiguous, near-even split of n images: the first n % world ranks get one extra."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_sum_(tensors, group=None):
    """In-place SUM all-reduce (no averaging: yolo_loss is a batch sum)."""
    if world(group)[1] == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors


class GradBucket:
    """The training step's one exchange: Dense weight gradient, bias gradient and the batch loss live in ONE
    contiguous fp32 buffer [dWt (rows x ldk) | db (units) | loss (1)], so a step issues a single SUM all-reduce
    (73.9 MB at rows = 1472, ldk = 12544).  `dwt`, `db`, `loss` are views for the kernels to write into; `grads` is the
    part Adam consumes in one launch (the parameters are kept in the same [Wt | b] order, engine.Trainer)."""

    def __init__(self, rows, ldk, units, device):
        self.n_w, self.units = rows * ldk, units
        self.flat = torch.zeros(self.n_w + units + 1, dtype=torch.float32, device=device)
        self.dwt = self.flat[:self.n_w].view(rows, ldk)
        self.db = self.flat[self.n_w:self.n_w + units]
        self.loss = self.flat[self.n_w + units:]
        self.grads = self.flat[:self.n_w + units]
        self.always = False     # developer rehearsal: issue the collective on a one-rank group too

    def nbytes(self):
        return self.flat.numel() * 4

    def allreduce(self, group=None, async_op=False):
        """In-place SUM over the data-parallel group, no division (yolo_loss is a batch sum: SURVEY Q12).
        -> the Work handle when async_op (None for a single rank)."""
        if world(group)[1] == 1 and not self.always:
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def all_gather_rows(out, inp, group=None, async_op=False):
    """out (world * n, ...) <- every rank's inp (n, ...), in rank order (one all_gather_into_tensor: equal n on every rank).
    -> the Work handle when async_op and the backend runs the collective asynchronously on the device ('nccl' = RCCL), else
    None: the gloo rehearsal / test backend gathers device tensors through the host, synchronously."""
    if inp.is_cuda and dist.get_backend(group) != 'nccl':
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu(), group=group)
        out.copy_(host)
        return None
    work = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    return work if async_op else None


RECORD_WIDTH = 6        # x0, y0, x1, y1, score, class (-1 where the slot is not kept)


def pack_detections(boxes, score, cls, keep):
    """-> (n, slots, 6) float64 fixed-size records (float64 keeps the box corners bit-exact).
    Device tensors go through one HIP kernel (ty1_pack_detections); host tensors (gloo tests) through torch."""
    if boxes.is_cuda:
        from . import ops
        return ops.pack_detections(boxes, score, cls, keep)
    rec = torch.empty(boxes.shape[:2] + (RECORD_WIDTH,), dtype=torch.float64, device=boxes.device)
    rec[..., :4] = boxes
    rec[..., 4] = score
    rec[..., 5] = torch.where(keep.bool(), cls, torch.full_like(cls, -1))
    return rec


def unpack_detections(rec):
    keep = rec[..., 5] >= 0
    return rec[..., :4], rec[..., 4].float(), rec[..., 5].clamp(min=0).int(), keep.to(torch.uint8)


def gather_detections(rec, counts=None, group=None):
    """All-gather per-rank records into the global batch order.  `counts` = per-rank image counts
    (needed only when shards are ragged; equal shards use one all_gather_into_tensor)."""
    rank, ws = world(group)
    if ws == 1:
        return rec
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((ws * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
        return out
    cap = max(counts)
    padded = torch.zeros((cap,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    padded[:rec.shape[0]] = rec
    out = torch.empty((ws * cap,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * cap:r * cap + counts[r]] for r in range(ws)], dim=0)
