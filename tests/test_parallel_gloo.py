"""CPU, world_size 2, gloo: the data-parallel logic of yolo_amd.parallel (the same code RCCL runs
on the GPUs).  Compute is supplied by the oracle here because no GPU is present; what is under
test is the sharding, the SUM all-reduce (no division: SURVEY Q12) and the detection gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolo_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make_batch(N=8, C=20):
    rng = np.random.default_rng(7)
    from oracle import model_ref
    y_true = np.zeros((N, 7, 7, C + 5), np.float32)
    for n in range(N):
        for _ in range(3):
            r, c, k = rng.integers(0, 7), rng.integers(0, 7), rng.integers(0, C)
            if y_true[n, r, c, C + 4] == 0:
                y_true[n, r, c, k] = 1
                y_true[n, r, c, C:C + 4] = rng.random(4)
                y_true[n, r, c, C + 4] = 1
    feats = rng.standard_normal((N, 64)).astype(np.float32)
    w = (0.3 * rng.standard_normal((64, 1470))).astype(np.float32)
    return y_true, feats, w


def _local_grads(y_true, feats, w):
    """Oracle forward/backward of Dense+head+loss on one shard -> (dW, loss)."""
    from oracle.loss_ref import yolo_loss_ref
    f = torch.from_numpy(feats)
    wt = torch.from_numpy(w).requires_grad_(True)
    z = f @ wt
    n = z.shape[0]
    y = torch.cat([torch.softmax(z[:, :980].reshape(n, 7, 7, 20), -1), torch.sigmoid(z[:, 980:1078].reshape(n, 7, 7, 2)),
                   torch.sigmoid(z[:, 1078:].reshape(n, 7, 7, 8))], -1)
    loss = yolo_loss_ref(torch.from_numpy(y_true), y)
    loss.backward()
    return wt.grad, loss.detach().reshape(1), y.detach().numpy()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import oracle
        y_true, feats, w = _make_batch()
        N = y_true.shape[0]
        # ragged split on purpose: 5 + 3 images
        bounds = [(0, 5), (5, 8)]
        lo, hi = bounds[rank]
        dw, loss, y = _local_grads(y_true[lo:hi], feats[lo:hi], w)
        # the training step's exchange: [dWt | db | loss] packed in one bucket = ONE collective (engine.Trainer uses
        # this same class on the GPU); count the calls to prove it
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1]
        bucket = parallel.GradBucket(1472, 64, 1470, 'cpu')
        bucket.dwt[:1470].copy_(dw.t())
        bucket.db.copy_(dw.sum(0))                       # any per-unit vector: the slot must travel and sum too
        bucket.loss.copy_(loss)
        db_local = bucket.db.clone()
        work = bucket.allreduce(async_op=True)
        work.wait()
        dist.all_reduce = real
        ok_one = calls == [1472 * 64 + 1470 + 1]
        dw, loss = bucket.dwt[:1470].t().contiguous(), bucket.loss.clone()
        other = [torch.zeros_like(db_local) for _ in range(world)]
        dist.all_gather(other, db_local)
        ok_one = ok_one and torch.allclose(bucket.db, other[0] + other[1]) and not bucket.dwt[1470:].any()
        extra = [torch.ones(3) * (rank + 1)]
        parallel.allreduce_sum_(extra)                   # the generic helper still sums without dividing
        ok_one = ok_one and bool((extra[0] == 3).all())
        dw_full, loss_full, y_full = _local_grads(y_true, feats, w)
        ok_grad = torch.allclose(dw, dw_full, rtol=1e-4, atol=1e-6)
        ok_loss = abs(loss.item() - loss_full.item()) <= 1e-5 * abs(loss_full.item())
        # detection gather: per-rank records -> global batch order, bit-exact
        det = oracle.decode_nms(y, 2, 20, 448)
        rec = parallel.pack_detections(*(torch.from_numpy(det[k]) for k in ('boxes', 'score', 'cls', 'keep')))
        allrec = parallel.gather_detections(rec, counts=[b - a for a, b in bounds])
        full = oracle.decode_nms(y_full, 2, 20, 448)
        boxes, score, cls, keep = parallel.unpack_detections(allrec)
        ok_det = (np.array_equal(keep.numpy(), full['keep']) and
                  np.array_equal(boxes.numpy().view(np.uint64), full['boxes'].view(np.uint64)) and
                  np.array_equal(cls.numpy()[full['keep'] == 1], full['cls'][full['keep'] == 1]))
        # equal shards take the single-collective path
        eq = parallel.gather_detections(rec[:3].contiguous())
        ok_eq = eq.shape[0] == 6
        spans = [parallel.shard_bounds(N, r, world) for r in range(world)]
        q.put((rank, ok_grad, ok_loss, ok_det, ok_eq, spans == [(0, 4), (4, 8)], ok_one))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_detection_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, *oks in results:
        assert all(oks), (rank, oks)
