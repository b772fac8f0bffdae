"""GPU, two ranks on ONE card: engine.Trainer.step with a real process group of world size 2 (the group plumbing, the
single [dWt | db | loss] collective, the side-stream overlap) on the HIP kernels.  RCCL refuses two ranks on one device, so
the collective travels over gloo here (it accepts device tensors); with RCCL the same code path is covered on one rank by
test_gpu_api.py::test_overlapped_step_equals_serial_step_one_rank_rccl.  What is checked: the data-parallel step on ragged
shards (5 + 3 images) gives the full batch's summed loss and gradient (SUM, no division: SURVEY Q12), and both ranks
hold bit-identical weights afterwards; the same for Trainer.exchange = 'gather' on equal shards (4 + 4), where features and
dz travel instead of the weight gradient."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from test_gpu_api import _labels
        from yolo_amd import params as P
        from yolo_amd.engine import DeviceModel, Trainer
        dev = 'cuda:0'
        torch.cuda.set_device(0)
        solo = [dist.new_group([r]) for r in range(world)][rank]          # (every rank takes part in creating every group)
        N = 8
        prm = P.make_random_params(seed=8, dense_init='glorot', bn='random')
        x = torch.rand((N, 448, 448, 3), generator=torch.Generator().manual_seed(11)).to(dev)
        y = torch.from_numpy(_labels(N, 20, 12)).to(dev)
        lo, hi = [(0, 5), (5, 8)][rank]
        # reference: the whole batch in one process (a one-rank group: no collective)
        ref_model = DeviceModel(prm, device=dev)
        ref = Trainer(ref_model, lr=1e-5, group=solo, overlap=False)
        loss_full = ref.step(x, y).item()
        g_full = ref.bucket.grads.clone()
        # data parallel: this rank's shard, default group (world 2), exchange + Adam on the side stream
        model = DeviceModel(prm, device=dev)
        calls = []
        real = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        dist.all_reduce = counting
        tr = Trainer(model, lr=1e-5, overlap=True)
        loss_dp = tr.step(x[lo:hi].contiguous(), y[lo:hi].contiguous(), wait=False)
        tr.join()
        torch.cuda.synchronize()
        dist.all_reduce = real
        ok_one = calls == [tr.bucket.flat.numel()]
        ok_loss = abs(loss_dp.item() - loss_full) <= 2e-4 * abs(loss_full)
        # the summed gradient: shards run their own forward passes (other K-split plans -> last-bit differences in the
        # features), so a tolerance, not bits
        g = tr.bucket.grads
        scale = g_full.abs().max().item()
        ok_grad = (g - g_full).abs().max().item() <= 2e-3 * scale
        # a second step keeps working (the update of step 1 is consumed in front of step 2's Dense forward)
        loss2 = tr.step(x[lo:hi].contiguous(), y[lo:hi].contiguous(), wait=True).item()
        w = tr.pflat.cpu()
        ws = [torch.empty_like(w) for _ in range(world)] if rank == 0 else None
        dist.gather(w, ws, dst=0)
        ok_same = True if rank != 0 else bool(torch.equal(ws[0], ws[1]))
        # Trainer.exchange = 'gather': equal shards (4 + 4), features and dz all-gathered, every rank computes the group's
        # weight gradient itself - no all-reduce; against the same whole-batch reference
        lo, hi = [(0, 4), (4, 8)][rank]
        model_g = DeviceModel(prm, device=dev)
        tg = Trainer(model_g, lr=1e-5, overlap=True)
        tg.exchange = 'gather'
        calls.clear()
        dist.all_reduce = counting
        loss_g = tg.step(x[lo:hi].contiguous(), y[lo:hi].contiguous(), wait=False)
        tg.join()
        torch.cuda.synchronize()
        dist.all_reduce = real
        ok_gather = (not calls and abs(loss_g.item() - loss_full) <= 2e-4 * abs(loss_full)
                     and (tg.bucket.grads - g_full).abs().max().item() <= 2e-3 * scale)
        wg = tg.pflat.cpu()
        wgs = [torch.empty_like(wg) for _ in range(world)] if rank == 0 else None
        dist.gather(wg, wgs, dst=0)
        ok_gather = ok_gather and (True if rank != 0 else bool(torch.equal(wgs[0], wgs[1])))
        q.put((rank, ok_one, ok_loss, ok_grad, ok_same, bool(np.isfinite(loss2)), ok_gather, loss_dp.item(), loss_full))
    finally:
        dist.destroy_process_group()


def test_trainer_step_world_size_two_on_one_gpu():
    from test_parallel_gloo import _free_port
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, *oks, l_dp, l_full in results:
        assert all(oks), (rank, oks, l_dp, l_full)
