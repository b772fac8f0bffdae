"""The compiled hot path: frozen conv backbone -> Dense -> head activations [-> decode + NMS], run
as a fixed sequence of libty1.so kernel launches on one HIP stream with all buffers preallocated.

Replaces what Keras' Model.predict executes for /root/reference/models/model_tiny_yolov1.py:42-99
and the NumPy post-processing of /root/reference/tiny_yolov1.py:82-140.
"""
import numpy as np
import torch

from . import ops
from .params import CONV_CHANNELS, N_POOL, BN_EPS, LEAKY_ALPHA, hwio_to_ohwi, dense_to_device_layout


class DeviceModel:
    def __init__(self, params, S=7, B=2, C=20, img_size=448, device='cuda:0', conv_specs=None, dtype='f32'):
        """dtype 'f32': the reference's arithmetic.  dtype 'bf16' (BASELINE config 3): conv and Dense operands (image, weights,
        activations) in bf16 with fp32 accumulation; BN, bias, head activations and decode/NMS stay fp32."""
        if dtype not in ('f32', 'bf16'):
            raise ValueError(f"dtype must be 'f32' or 'bf16', got {dtype!r}")
        self.dtype = dtype
        self.S, self.B, self.C, self.img_size = S, B, C, img_size
        self.device = torch.device(device)
        self.units = S * S * (C + 5 * B)
        # (cin, cout, pool) per conv block; default = cfg/yolov1-tiny.cfg as transcribed by the reference
        self.specs = conv_specs or [(ci, co, i < N_POOL) for i, (ci, co) in enumerate(CONV_CHANNELS)]
        self.conv = []
        for i, (cin, cout, pool) in enumerate(self.specs):
            scale, shift = ops.fold_bn(params[f'bn{i}_gamma'], params[f'bn{i}_beta'], params[f'bn{i}_mean'],
                                       params[f'bn{i}_var'], BN_EPS)
            w = torch.from_numpy(hwio_to_ohwi(params[f'conv{i}_w'])).to(self.device)
            if dtype == 'bf16' and cin != 3:
                w = ops.cast_bf16(w)
            self.conv.append(dict(
                w=w,
                scale=torch.from_numpy(scale).to(self.device), shift=torch.from_numpy(shift).to(self.device),
                pool=pool))
        assert params['dense_w'].shape[1] == self.units, (params['dense_w'].shape, self.units)
        self.dense_wt = torch.from_numpy(dense_to_device_layout(params['dense_w'])).to(self.device)
        self.dense_b = torch.from_numpy(np.ascontiguousarray(params['dense_b'], np.float32)).to(self.device)
        self.K = self.dense_wt.shape[1]
        self.dense_wt_bf16 = ops.cast_bf16(self.dense_wt) if dtype == 'bf16' else None
        self._ws = {}
        self._graphs, self._inputs = {}, {}
        self.fused01 = True      # bf16: conv blocks 0 and 1 in one launch (ty1_conv3x3_x2_c3_bf16); False: two launches (A/B, tests)

    # ------------------------------------------------------------------ workspaces
    def workspace(self, N):
        ws = self._ws.get(N)
        if ws is None:
            dev, f32 = self.device, torch.float32
            acts, h, scratch = [], self.img_size, 0
            bf = self.dtype == 'bf16'
            for i, (cin, cout, pool) in enumerate(self.specs):
                need = ops.conv3x3_workspace_bytes_bf16 if bf else ops.conv3x3_workspace_bytes
                scratch = max(scratch, need(N, h, h, cin, cout, pool))
                if pool:
                    h //= 2
                acts.append(torch.empty((N, h, h, cout), dtype=torch.bfloat16 if bf else f32, device=dev))
            splits = ops.pick_splits(self.K, N, self.units, chunk=64 if bf else ops.DENSE_K_ALIGN)
            ldz = ops.round_up(self.units, 4)
            slots = self.S * self.S * self.B
            ws = dict(acts=acts, scratch=torch.empty((max(scratch, 16),), dtype=torch.uint8, device=dev), part=torch.empty((splits, N, ldz), dtype=f32, device=dev),
                      y=torch.empty((N, self.S, self.S, self.C + 5 * self.B), dtype=f32, device=dev),
                      det=(torch.empty((N, slots, 4), dtype=torch.float64, device=dev),
                           torch.empty((N, slots), dtype=f32, device=dev),
                           torch.empty((N, slots), dtype=torch.int32, device=dev),
                           torch.empty((N, slots), dtype=torch.uint8, device=dev)))
            self._ws[N] = ws
        return ws

    # ------------------------------------------------------------------ forward pieces
    def backbone(self, x, ws=None, after_layer=None):
        """x (N,H,W,3) f32 NHWC on the device -> features (N, 12544) (HWC flatten).
        after_layer: optional {index: callable} - a callable runs on the host once layer `index` has been enqueued (the
        trainer uses it to start the deferred Adam launch under the compute-bound layers and to take the feature buffer
        back from the tail in flight before the last layer writes it)."""
        ws = ws or self.workspace(x.shape[0])
        fused01 = self.fused01 and self._fusable01(x)
        for i, (layer, out) in enumerate(zip(self.conv, ws['acts'])):
            if after_layer and i - 1 in after_layer:
                after_layer[i - 1]()
            if fused01 and i < 2:          # blocks 0 and 1 of the bf16 network are one launch: the tensor between them stays in LDS
                if i == 1:
                    l0 = self.conv[0]
                    x = ops.conv3x3_x2_c3_bf16(x, l0['w'], l0['scale'], l0['shift'], layer['w'], layer['scale'], layer['shift'],
                                               LEAKY_ALPHA, out=out)
                continue
            if self.dtype == 'bf16':
                x = ops.conv3x3_bn_leaky_bf16(x, layer['w'], layer['scale'], layer['shift'], layer['pool'], LEAKY_ALPHA,
                                              out=out, workspace=ws['scratch'])
            else:
                x = ops.conv3x3_bn_leaky(x, layer['w'], layer['scale'], layer['shift'], layer['pool'], LEAKY_ALPHA, out=out,
                                         workspace=ws['scratch'])
        return x.reshape(x.shape[0], -1)

    def _fusable01(self, x):
        s = self.specs
        return (self.dtype == 'bf16' and len(s) >= 2 and tuple(s[0]) == (3, 16, True) and tuple(s[1]) == (16, 32, True)
                and x.shape[1] % 4 == 0 and x.shape[2] % 4 == 0 and 0.0 < LEAKY_ALPHA <= 1.0)

    def head(self, feats, ws=None):
        ws = ws or self.workspace(feats.shape[0])
        wt = self.dense_wt_bf16 if self.dtype == 'bf16' else self.dense_wt
        part = ops.dense_splitk(feats, wt, self.units, part=ws['part'])
        return ops.head_activate(part, self.dense_b, self.S, self.B, self.C, out=ws['y'])

    def predict(self, x):
        """-> y_pred (N,S,S,C+5B) on the device (valid until the next call with the same N).
        With a Trainer attached: an update left in flight by step(wait=False) must be joined first (Trainer.join()),
        otherwise this forward reads the weights from before it.  The same holds for detect()."""
        ws = self.workspace(x.shape[0])
        return self.head(self.backbone(x, ws), ws)

    FUSED_TAIL_MIN_BATCH = 1       # (with 85 Dense splits the two-launch tail won below 16 images; with 21 the fused one wins from 1)

    def detect(self, x, iou_thr=0.2, want_y=False):
        """-> (boxes f64 (N,S*S*B,4), score, cls, keep) on the device.  From FUSED_TAIL_MIN_BATCH images on, everything
        behind the Dense GEMM is one launch (ty1_head_decode_nms_f32) and the head tensor stays in LDS unless want_y."""
        N = x.shape[0]
        ws = self.workspace(N)
        feats = self.backbone(x, ws)
        if N < self.FUSED_TAIL_MIN_BATCH:
            return ops.decode_nms(self.head(feats, ws), self.B, self.C, self.img_size, iou_thr, out=ws['det'])
        wt = self.dense_wt_bf16 if self.dtype == 'bf16' else self.dense_wt
        part = ops.dense_splitk(feats, wt, self.units, part=ws['part'])
        return ops.head_decode_nms(part, self.dense_b, self.S, self.B, self.C, self.img_size, iou_thr,
                                   y=ws['y'] if want_y else None, out=ws['det'])


    def detect_u8(self, frames_u8, swap_rb=False, iou_thr=0.2):
        """Equal-sized uint8 frames (n,H0,W0,3) on the device -> detect(): cv2.resize + [BGR->RGB] + /255 of the whole batch
        in one launch (ty1_preprocess_batch_u8; /root/reference/utils.py:11-15 over the batch loop of
        data_sequence.py:94-99) into the batch's input buffer, then the network.  uint8 frames are a quarter of the fp32
        batch on the PCIe link: from host frames a 64-image step stays compute-bound (bench.py, "from_host_frames")."""
        x = ops.preprocess_batch_u8(frames_u8, (self.img_size, self.img_size), swap_rb, out=self.input_buffer(frames_u8.shape[0]))
        return self.detect(x, iou_thr)

    # ------------------------------------------------------------------ one HIP-graph launch per batch
    def detect_replay(self, x, iou_thr=0.2):
        """detect() replayed as ONE HIP graph: the fixed launch sequence of a batch size (8 conv blocks, fix-ups, Dense,
        head + decode + NMS) is captured once and then costs one host call instead of a dozen.  At batch 1 - the reference's
        own use, tiny_yolov1.py:142-160 detect(image) per frame - the eager sequence is bound by the host's launch calls, not
        by the GPU.  Same kernels, same buffers, same bits as detect().

        x is copied into the graph's input buffer unless it IS that buffer (input_buffer(N): write the batch there and
        pass it to skip the copy).  The returned tensors are the graph's outputs: valid until the next replay with this N."""
        N = x.shape[0]
        key = (N, float(iou_thr))
        ent = self._graphs.get(key)
        if ent is None:
            xin = self.input_buffer(N)
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):          # first calls allocate the workspace: outside the capture
                self.detect(xin, iou_thr)
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.detect(xin, iou_thr)
            ent = self._graphs[key] = (graph, out)
        xin = self.input_buffer(N)
        if x.data_ptr() != xin.data_ptr():
            xin.copy_(x)
        ent[0].replay()
        return ent[1]

    def input_buffer(self, N):
        """The (N, img, img, 3) fp32 batch the graphs of batch size N read."""
        buf = self._inputs.get(N)
        if buf is None:
            buf = self._inputs[N] = torch.zeros((N, self.img_size, self.img_size, 3), dtype=torch.float32, device=self.device)
        return buf


class HostFramePipeline:
    """Detection from frames in HOST memory (the per-frame loop of /root/reference/tiny_yolov1.py:142-160 and the batch loop of
    data_sequence.py:94-99, fed the way a server is: frames arrive in host buffers).  submit(frames) copies one batch of
    equal-sized uint8 frames (n,H0,W0,3) - pinned host memory for an asynchronous copy - to one of two device buffers on a
    copy stream, so the copy of batch i + 1 runs under the network of batch i, then resizes the whole batch in one launch
    (ty1_preprocess_batch_u8) and runs detect().  uint8 frames are a quarter of the fp32 batch on the link.
    The host stays at most two batches ahead: submit() waits (on the host) for the device buffer it is about to overwrite -
    left to run fifty batches ahead, hipMemcpyAsync blocks at times while the pipeline is still empty - and for the copy of
    the PREVIOUS batch: when submit() returns, that batch's host buffer may be refilled (two host buffers, used in turn,
    are enough)."""

    def __init__(self, model, n, H0, W0, swap_rb=False, iou_thr=0.2):
        self.model, self.n, self.swap_rb, self.iou_thr = model, n, swap_rb, iou_thr
        dev = model.device
        self.frames = [torch.empty((n, H0, W0, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.copy = torch.cuda.Stream(dev)
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.free = [torch.cuda.Event() for _ in range(2)]
        self.i = 0

    def submit(self, frames_u8_host):
        """-> detect()'s (boxes, score, cls, keep) for this batch, on the device; valid until the next submit()."""
        if tuple(frames_u8_host.shape) != tuple(self.frames[0].shape) or frames_u8_host.dtype != torch.uint8:
            raise ValueError(f'frames: expected uint8 {tuple(self.frames[0].shape)}, got {frames_u8_host.dtype} {tuple(frames_u8_host.shape)}')
        m, b = self.model, self.i & 1
        if self.i >= 2:
            self.free[b].synchronize()               # the preprocess of two batches ago has read this device buffer
        if self.i >= 1:
            self.ready[b ^ 1].synchronize()          # the previous batch has left its host buffer
        self.i += 1
        cur = torch.cuda.current_stream(m.device)
        with torch.cuda.stream(self.copy):
            self.frames[b].copy_(frames_u8_host, non_blocking=True)
            self.ready[b].record(self.copy)
        cur.wait_event(self.ready[b])
        x = ops.preprocess_batch_u8(self.frames[b], (m.img_size, m.img_size), self.swap_rb, out=m.input_buffer(self.n))
        self.free[b].record(cur)
        return m.detect(x, self.iou_thr)


class Trainer:
    """One training step of the reference's configuration (train.py:59-60): the frozen
    backbone and the Dense layer run forward, yolo_loss gives dL/dy_pred in closed form, the head
    Jacobian and the Dense weight gradient follow, gradients are all-reduced with SUM over the
    data-parallel group (the loss is a batch sum: SURVEY Q12, no division by world size), and Adam
    (Keras defaults) updates the only trainable parameters: the Dense kernel and bias (Q8).

    Exchange and update are built to scale: [dWt | db | loss] is one flat buffer (parallel.GradBucket) = ONE all-reduce
    per step, the parameters and Adam moments use the same [Wt | b] order = ONE Adam launch.  With overlap=True the
    whole trainable tail - Dense forward, head + loss + head Jacobian, weight gradient, all-reduce, Adam - runs on a
    side stream, in order, while the compute stream already runs the NEXT batch's backbone: legal because every conv
    and BatchNormalization layer is frozen (/root/reference/models/model_tiny_yolov1.py:43-87, trainable=False), so
    the backbone reads nothing the tail writes, and the side stream's order alone puts step i+1's Dense forward behind
    step i's update.  The only edge back: the next backbone's last layer overwrites the feature buffer, so the compute
    stream waits (right before that layer, long after the fact) for the tail to have read it.  step(wait=True),
    the default, ends with join(): loss and weights are then ordered on the caller's stream.  The arithmetic and its
    order are those of the serial step: same bits (tests/test_gpu_api.py)."""

    UPDATE_AFTER_LAYER = 2        # 112x112x32->64: in-call A/B tools/experiments/tail_side_ab.sh at 32 images (ms per step): at once 0.977, after layer 0 / 1 / 2 / 3 / 4 / 5: 0.979 / 1.004 / 0.971 / 0.977 / 0.973 / 0.988

    def __init__(self, model, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, group=None, overlap=True):
        if model.dtype != 'f32':
            raise ValueError("training runs in fp32 (the reference's arithmetic): build the model with dtype='f32'")
        from . import parallel
        self.model, self.group = model, group
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        dev = self.device = model.device
        rows, ldk = model.dense_wt.shape
        self.bucket = parallel.GradBucket(rows, ldk, model.units, dev)
        self.dwt, self.db = self.bucket.dwt, self.bucket.db
        # re-home the trainable parameters into one flat buffer in the bucket's order; the model keeps using views
        n_w = rows * ldk
        self.pflat = torch.empty(n_w + model.units, dtype=torch.float32, device=dev)
        self.pflat[:n_w].view(rows, ldk).copy_(model.dense_wt)
        self.pflat[n_w:].copy_(model.dense_b)
        model.dense_wt, model.dense_b = self.pflat[:n_w].view(rows, ldk), self.pflat[n_w:]
        model._graphs.clear()            # captured graphs hold the old parameter addresses
        self.m, self.v = torch.zeros_like(self.pflat), torch.zeros_like(self.pflat)
        self.t = 0
        self.ldz = ops.round_up(model.units, 4)
        self._dz = {}
        self.side = torch.cuda.Stream(dev) if overlap else None
        self._feats_ready = torch.cuda.Event()   # compute stream: the backbone's features are written
        self._feats_free = torch.cuda.Event()    # side stream: Dense forward and weight gradient have read them
        self._y = {}                             # private label buffers of step(wait=False), by shape
        self._updated = torch.cuda.Event()
        self._pending = False            # an update is in flight on the side stream
        self.collective_events = None    # (start, end) CUDA events around the last all-reduce when time_collective is set
        self.time_collective = False
        self.skip_collective = False     # measurement only (bench: step time without the exchange)
        # How the ranks of a data-parallel group arrive at the same summed gradient:
        #  'allreduce': each rank's dWt = dz^T x, then ONE SUM all-reduce of [dWt | db | loss] (73.9 MB); any shard sizes.
        #  'gather':    all-gather of the features (n x 12544) and of dz (n x 1472, the local loss rides in a padding column),
        #               then every rank computes the GLOBAL dWt = dz_all^T x_all itself: 1.8 MB per rank on the links instead
        #               of 73.9 MB through a ring (xGMI is point to point: an all-reduce of that size is link-bound), for a
        #               weight-gradient GEMM world-size times deeper (its 73.8 MB store bounds it up to ~4 ranks, then the
        #               MFMAs: 60 us at 8).  Needs the same number of images on every rank.  Same sum, other order; every
        #               rank runs the same kernel on the same gathered rows: bit-identical weights across ranks.
        self.exchange = 'allreduce'
        self._gathered = {}
        self.tail_on_side = True         # False: Dense forward / loss / weight gradient on the compute stream (same bits; A/B)
        self.fused_tail = True           # False: the three separate launches (same bits; kept for the comparison test)
        # step(wait=False) only: the Adam launch of the update in flight is held back until the NEXT batch's backbone has
        # enqueued this layer (None: it follows the all-reduce at once).  The first two layers are bound by their own
        # instruction streams and by HBM, and the 516 MB Adam pass next to them costs them more than it hides; under the
        # MFMA-bound layers behind them it costs the step 30 of its 85 us (tools/experiments/tail_knockouts.sh).  The
        # all-reduce itself always starts at once.
        self.update_after_layer = Trainer.UPDATE_AFTER_LAYER
        self._adam_deferred = False
        # CUs the NEXT batch's persistent conv grids leave free while this step's exchange is in flight (ty1_set_reserved_cus):
        # somewhere for RCCL's workgroups to run.  0 = fill the chip (the default until an 8-GPU run says otherwise);
        # results do not depend on it.
        self.reserve_cus = 0

    def gradients(self, x, y_true):
        """Forward + backward on this rank's shard: fills self.dwt / self.db, returns the local loss (1,)."""
        return self.gradients_from_features(self.model.backbone(x, self.model.workspace(x.shape[0])), y_true)

    def gradients_from_features(self, feats, y_true):
        """The trainable part of the step (Dense -> head -> yolo_loss and back) on frozen-backbone features (N, K).
        Writes dWt, db and the local loss straight into the all-reduce bucket and returns the bucket's loss slot: a VIEW
        that the next call (or the next collective) overwrites - clone it to keep it.  An update still in flight on the
        side stream (step(wait=False)) is joined first: it reads the bucket this call writes."""
        self.join()
        return self._tail(feats, y_true)

    def _gathers(self):
        """True when this step exchanges features and dz instead of the weight gradient (self.exchange == 'gather' in a group
        of more than one rank, or the one-rank rehearsal of it)."""
        from . import parallel
        return (self.exchange == 'gather' and not self.skip_collective and self.ldz > self.model.units
                and (parallel.world(self.group)[1] > 1 or self.bucket.always))

    def _tail(self, feats, y_true):
        """Dense forward, head + loss + head Jacobian, Dense weight gradient: enqueued on the CURRENT stream.  In 'gather'
        mode the two all-gathers sit in between and the weight gradient is the group's (no all-reduce follows)."""
        from . import parallel
        m = self.model
        N = feats.shape[0]
        ws = m.workspace(N)
        dz = self._dz.get(N)
        if dz is None:
            dz = self._dz[N] = torch.empty((N, self.ldz), dtype=torch.float32, device=self.device)
        gathers = self._gathers()
        works = []
        if gathers:
            W = parallel.world(self.group)[1]
            bufs = self._gathered.get(N)
            if bufs is None:
                bufs = self._gathered[N] = (torch.empty((W * N, feats.shape[1]), dtype=torch.float32, device=self.device),
                                            torch.empty((W * N, self.ldz), dtype=torch.float32, device=self.device))
            feats_all, dz_all = bufs
            if self.time_collective:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            # the features leave at once: their gather runs under the Dense forward and the loss
            works.append(parallel.all_gather_rows(feats_all, feats, self.group, async_op=True))
        # the local loss: into the bucket, or - gather mode - into the first padding column of dz's first row (the kernels
        # write dz's padding, zeros, before the loss sum lands there; the weight gradient never reads it)
        loss_slot = dz.view(-1)[m.units:m.units + 1] if gathers else self.bucket.loss
        part = ops.dense_splitk(feats, m.dense_wt, m.units, part=ws['part'])
        if self.fused_tail:
            # head activation, loss forward / backward and head Jacobian in one launch; y and dL/dy stay in LDS
            ops.train_tail(part, m.dense_b, y_true, self.ldz, m.S, m.B, m.C, m.img_size, dz=dz, loss_out=loss_slot)
        else:
            y = ops.head_activate(part, m.dense_b, m.S, m.B, m.C, out=ws['y'])
            _, _, grad = ops.yolo_loss_fwd_bwd(y_true, y, m.B, m.C, m.img_size, loss_out=loss_slot)
            ops.head_activate_bwd(y, grad, self.ldz, m.B, m.C, out=dz)
        if not gathers:
            ops.dense_wgrad(dz, feats, m.units, self.dwt, self.db)
            return self.bucket.loss
        works.append(parallel.all_gather_rows(dz_all, dz, self.group, async_op=True))
        for w in works:
            if w is not None:
                w.wait()                     # nccl: the current stream waits for the collective, the host goes on
        if self.time_collective:
            e1.record()
            self.collective_events = (e0, e1)
        ops.dense_wgrad(dz_all, feats_all, m.units, self.dwt, self.db)
        # the group's loss: the ranks' local losses in rank order (the same on every rank)
        torch.sum(dz_all.view(W, N * self.ldz)[:, m.units], dim=0, keepdim=True, out=self.bucket.loss)
        return self.bucket.loss

    def _take_features_back(self):
        """The tail in flight reads the feature buffer the backbone's LAST layer writes: the compute stream waits for it
        to have done so (enqueued right before that layer: by then the tail is long past it)."""
        if self._pending:
            torch.cuda.current_stream(self.device).wait_event(self._feats_free)

    def _launch_deferred_adam(self):
        """The held-back Adam launch: on the side stream, behind the all-reduce already there and behind whatever the
        compute stream has enqueued so far."""
        if not self._adam_deferred:
            return
        self._adam_deferred = False
        cur = torch.cuda.current_stream(self.device)
        gate = torch.cuda.Event()
        gate.record(cur)
        with torch.cuda.stream(self.side):
            self.side.wait_event(gate)
            self._adam()
            self._updated.record(self.side)

    def join(self):
        """Make the current stream wait for the update in flight on the side stream (no host synchronisation)."""
        self._launch_deferred_adam()
        if self._pending:
            torch.cuda.current_stream(self.device).wait_event(self._updated)
            self._pending = False

    def _adam(self):
        self.t += 1
        ops.adam_step(self.pflat, self.bucket.grads, self.m, self.v, self.t, self.lr, self.beta1, self.beta2, self.eps)

    def _exchange_and_update(self, defer_adam=False, loss_out=None):
        """ONE all-reduce of the bucket, the global loss copied out (into loss_out (1,) if given), ONE Adam launch - on the
        current stream."""
        if not self.skip_collective and not self._gathers():      # ('gather': the exchange happened inside _tail)
            if self.time_collective:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            work = self.bucket.allreduce(self.group, async_op=True)
            if work is not None:
                # nccl (RCCL): makes the current stream wait for the collective's stream, the host goes on (that is what the
                # overlap is built on).  gloo (CPU tests, rehearsals): blocks the HOST until the exchange is done.
                work.wait()
            if self.time_collective:
                e1.record()
                self.collective_events = (e0, e1)
        loss = self.bucket.loss.clone() if loss_out is None else loss_out.copy_(self.bucket.loss)
        if defer_adam:
            self._adam_deferred = True
        else:
            self._adam()
        return loss

    def step(self, x, y_true, wait=True):
        """x (n,448,448,3), y_true (n,S,S,C+5) on the device -> global batch loss (1,) on the device.
        wait=False leaves everything behind the backbone - Dense forward, loss, gradients, exchange, update - in flight on
        the side stream, in order behind the previous step's update; the next step() puts its backbone on the compute stream
        at once.  The returned loss is then valid only after join() / a device synchronise; y_true is copied before the
        call returns, so the caller may refill it."""
        return self.steps(x, y_true, x.shape[0], wait)

    def steps(self, x, y_true, batch, wait=True):
        """k = n / batch consecutive optimisation steps on x (n,448,448,3), y_true (n,S,S,C+5): the FROZEN backbone runs once
        over all n images - it does not depend on anything the steps change, and its kernels run closer to their peak on 64
        or 128 images than on 32 - then each batch of `batch` images gets its own Dense forward, loss, gradient, exchange
        and Adam update, in order, exactly as k calls of step() would (the features of an image may differ from those of a
        smaller launch in the last bits: other K-split plans).  -> the k global batch losses (k,).  wait as in step()."""
        n, b = x.shape[0], int(batch)
        if b <= 0 or n % b or y_true.shape[0] != n:
            raise ValueError(f'steps: {n} images / {y_true.shape[0]} label rows do not split into batches of {b}')
        k = n // b
        part = lambda t, j: t[j * b:(j + 1) * b]
        if self.side is None:
            feats = self.model.backbone(x, self.model.workspace(n))
            losses = []
            for j in range(k):
                self._tail(part(feats, j), part(y_true, j))
                losses.append(self._exchange_and_update())
            return torch.cat(losses)
        cur = torch.cuda.current_stream(self.device)
        hooks = {}
        if self._pending:
            last = len(self.model.conv) - 2
            gate = last if self.update_after_layer is None else min(self.update_after_layer, last)
            hooks[gate] = self._launch_deferred_adam
            hooks[last] = (lambda: (self._take_features_back(), self._launch_deferred_adam())) if gate == last else self._take_features_back
        shrink = self.reserve_cus > 0 and self._pending
        if shrink:
            ops.set_reserved_cus(self.reserve_cus)
        try:
            feats = self.model.backbone(x, self.model.workspace(n), hooks)   # frozen: overlaps the tail + update in flight
        finally:
            if shrink:
                ops.set_reserved_cus(0)
        if self.tail_on_side and not wait:
            # the side stream reads the labels after this call has returned: a private copy (157 KB at 32 images), made
            # here - behind the wait for the previous tail, which read the same buffer
            yb = self._y.get(y_true.shape)
            if yb is None:
                yb = self._y[y_true.shape] = torch.empty_like(y_true)
            yb.copy_(y_true)
            y_true = yb
        losses = torch.empty((k,), dtype=torch.float32, device=self.device)    # written on the side stream, slot by slot
        losses.record_stream(self.side)
        for j in range(k):
            if not self.tail_on_side:          # round-2 order, kept for the A/B: the tail on the compute stream, behind the update
                self.join()
                self._tail(part(feats, j), part(y_true, j))
            self._feats_ready.record(cur)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self._feats_ready)
                if self.tail_on_side:
                    self._tail(part(feats, j), part(y_true, j))
                if j == k - 1:
                    self._feats_free.record(self.side)
                # only the last update can wait for the next backbone: the next Dense forward needs the others at once
                defer = j == k - 1 and not wait and self.update_after_layer is not None
                self._exchange_and_update(defer_adam=defer, loss_out=losses[j:j + 1])
                if not defer:
                    self._updated.record(self.side)
            self._pending = True
        if wait:
            self.join()
        return losses
