"""Whole-bag data parallelism over RCCL / xGMI: one process per GPU, B bags per rank, no tensor sharding.

Replaces the reference's ``DDP(model, find_unused_parameters=True, broadcast_buffers=False)`` wrap
(main.py:118-119) for this path:
  * gradients are averaged with bucketed asynchronous all-reduces launched from post-accumulate-grad hooks WHILE backward
    is still running (buckets follow reverse parameter order, i.e. classifier -> immune branch -> tumor branch -> omic
    encoders; `stats["launched_in_backward"]` counts them per step);
  * every reduction has completed, and ``.grad`` holds the averaged value, before ``backward()`` returns - the reference's
    train step reads ``model.module.classifier.weight.grad`` right after it (train_test.py:158);
  * parameters that receive no gradient (42 tensors with attn_dim = 2: attn1d.*, cls_token, _fc2, ...) are
    found in the FIRST step - the set is a property of the model, identical on every rank - and from then on they neither
    hold a bucket back nor travel: their slots stay zero, buckets made only of them are skipped, ``.grad`` stays None.
    No graph traversal (``find_unused_parameters``) is needed.  A parameter of that set that does get a gradient later
    raises (call ``reset_unused()`` after changing what the model computes);
  * the 1 / world factor is folded into the reduction (ReduceOp.AVG on RCCL; pre-scaled sum elsewhere): nothing runs
    between the collective's completion and the return of backward() but the wait itself;
  * ``.module`` exposes the wrapped model as DDP does; parameters and buffers are broadcast from rank 0 once, at wrap time;
  * SyncBatchNorm (main.py:118, only with fusion_type 'pofusion') issues its own all-gather / all-reduce inside forward and backward:
    they interleave with the bucket all-reduces in the order autograd reaches them, which is the same on every rank
    (tests/test_gpu_data_parallel.py::test_syncbn_pofusion_two_ranks).
The payload is ~4.65 MB per step for DeformPathomicNet (1.96 MB for one DeformCrossTransMIL), so the collectives are
latency-bound: the default bucket is 512 KiB - 4 to 9 buckets, enough for the first ones to be on the wire while the
position-bias backward (the longest kernel of the step) of the other branch still runs."""
from __future__ import annotations

import time
from typing import List

import torch
import torch.distributed as dist
from torch import nn


class _Bucket:
    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.offsets = []
        o = 0
        for p in params:
            self.offsets.append(o)
            o += p.numel()
        self.flat = None
        self.views = None
        self.used = None                      # per parameter: receives a gradient (known after the first step)
        self.pending = 0
        self.complete = False
        self.ready = [False] * len(params)
        self.work = None
        self.skip = False


class BagDataParallel(nn.Module):
    def __init__(self, module: nn.Module, bucket_bytes: int = 512 << 10, process_group=None, broadcast: bool = True,
                 collectives_at_world_1: bool = False):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # tests: run the whole reducer (hooks, buckets, the backend's all-reduce with its averaging op, the end-of-backward wait) in a ONE-rank group -
        # the only way to put RCCL under this code on a one-GPU box (RCCL refuses two ranks per device); the average over one rank is the identity
        self._reduce_at_1 = bool(collectives_at_world_1 and dist.is_initialized() and self.world == 1)
        params = [p for p in module.parameters() if p.requires_grad]
        if (self.world > 1 or self._reduce_at_1) and broadcast:
            # rank 0's state at wrap time, parameters AND buffers - as DDP's constructor does whatever `broadcast_buffers` says (that flag,
            # False at main.py:119, only stops the per-forward re-broadcast).  Buffers matter with fusion_type 'pofusion': main.py:118
            # converts BilinearFusion's BatchNorm1d to SyncBatchNorm, whose running statistics must start equal on every rank
            for p in module.parameters():
                dist.broadcast(p.data, src=0, group=process_group)
            for b in module.buffers():
                dist.broadcast(b.data, src=0, group=process_group)
        self._buckets: List[_Bucket] = []
        self._where = {}
        cur, cur_bytes = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
        if cur:
            self._buckets.append(_Bucket(cur))
        for bi, b in enumerate(self._buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
        self._armed = False
        self._known = False                   # the grad-less set has been recorded
        self._avg_op = None
        self.stats = {"buckets": len(self._buckets), "launched_in_backward": 0, "skipped": 0, "steps": 0, "hook_host_ms": 0.0,
                      "used_mismatch": 0}
        self._hook_s = 0.0                    # host time spent inside the gradient hooks of the current backward
        if self.world > 1 or self._reduce_at_1:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def timing(self):
        """Event times of the LAST backward (synchronises): `exposed_wait_ms` = GPU time between the end of the backward kernels and the
        completion of the last all-reduce on the compute stream (what the overlap did NOT hide), `allreduce_span_ms` = first bucket
        launch -> last completion (the collectives' wall span, most of it under the backward), and their quotient."""
        ev = getattr(self, "_ev", None)
        if not ev or "reduced" not in ev or ev.get("first_launch") is None:
            return {}
        torch.cuda.synchronize()
        span = ev["first_launch"].elapsed_time(ev["reduced"])
        exposed = ev["backward_end"].elapsed_time(ev["reduced"])
        return {"exposed_wait_ms": exposed, "allreduce_span_ms": span, "hidden_share": (1.0 - exposed / span) if span > 0 else None}

    def reset_unused(self):
        """Forget which parameters are grad-less (after the model's control flow changed); the next step re-learns the set."""
        self._known = False
        for b in self._buckets:
            b.used = None
            b.skip = False

    # ---- hooks ------------------------------------------------------------------------------
    def _arm(self):
        if not self._armed:
            self._armed = True
            self._hook_s = 0.0
            self._next = 0
            self._ev = {"first_launch": None}         # HIP events of this backward on the compute stream (timing())
            self.stats["launched_in_backward"] = 0
            for b in self._buckets:
                b.ready = [False] * len(b.params)
                b.work = None
                b.complete = False
                b.pending = len(b.params) if b.used is None else sum(b.used)
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _on_grad(self, p: nn.Parameter):
        t0 = time.perf_counter()
        try:
            self._on_grad_timed(p)
        finally:
            self._hook_s += time.perf_counter() - t0

    def _on_grad_timed(self, p: nn.Parameter):
        self._arm()
        bi, pi = self._where[p]
        b = self._buckets[bi]
        if b.used is not None and not b.used[pi]:
            raise RuntimeError("BagDataParallel: a parameter that received no gradient in the first step now has one; the set of "
                               "grad-less parameters must be static (call reset_unused() after changing the model's control flow)")
        if not b.ready[pi]:
            b.ready[pi] = True
            b.pending -= 1
            if b.pending == 0:
                b.complete = True
                # collectives must be issued in the SAME order on every rank: buckets are launched strictly in index order
                # (= the order their gradients become ready in the normal case); a complete bucket behind an incomplete one
                # waits for it - at the latest until the end-of-backward flush, which also walks the buckets in index order
                while self._next < len(self._buckets):
                    nb = self._buckets[self._next]
                    if nb.skip:
                        self._next += 1
                    elif nb.complete and nb.work is None:
                        self._launch(nb)
                        self.stats["launched_in_backward"] += 1
                        self._next += 1
                    else:
                        break

    def _reduce_op(self, t: torch.Tensor):
        """(op, pre-scale): ReduceOp.AVG where the backend has it (RCCL), else SUM of pre-scaled values."""
        if self._avg_op is None:
            backend = dist.get_backend(self.group)
            self._avg_op = (backend == "nccl")
        return (dist.ReduceOp.AVG, None) if self._avg_op else (dist.ReduceOp.SUM, 1.0 / self.world)

    def _launch(self, b: _Bucket):
        ref = b.params[0]
        if b.flat is None or b.flat.device != ref.device:
            b.flat = torch.zeros(b.numel, dtype=ref.dtype, device=ref.device)
            b.views = [b.flat[o:o + p.numel()].view_as(p) for p, o in zip(b.params, b.offsets)]
        # gather the gradients into the flat bucket with ONE multi-tensor copy (a copy kernel per parameter would add
        # ~100 launches per step); a gradient that already is the bucket's view (kept from the previous step) needs none
        live = [(v, p.grad) for v, p, r in zip(b.views, b.params, b.ready) if r and p.grad is not None]
        todo = [(v, g) for v, g in live if g.data_ptr() != v.data_ptr()]
        # slots without a gradient this step contribute zeros.  Slots of the static grad-less set were zeroed when the bucket
        # was created and are never written; any other missing slot (first step, or a used parameter that got nothing this
        # time) still holds the previous step's average and is cleared here.  Steady state (set known, every used slot ready):
        # nothing to clear, no per-slot work on the backward thread
        if b.used is None or len(live) != sum(b.used):
            for v, p, r, u in zip(b.views, b.params, b.ready, b.used if b.used is not None else [True] * len(b.params)):
                if u and not (r and p.grad is not None):
                    v.zero_()
        if todo:
            torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])
        op, scale = self._reduce_op(b.flat)
        if scale is not None:
            b.flat.mul_(scale)
        if b.flat.is_cuda and self._ev.get("first_launch") is None:
            self._ev["first_launch"] = torch.cuda.Event(enable_timing=True)
            self._ev["first_launch"].record()
        b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)

    def _finalize(self):
        # runs at the end of backward(): flush buckets that could not complete in the hooks (first step: grad-less parameters
        # are not known yet), wait, and hand every parameter the bucket's view as its .grad (no copy back, no arithmetic)
        on_gpu = self._buckets[0].params[0].is_cuda
        if on_gpu:                                    # the compute stream has every backward kernel queued: what follows it is exposed
            self._ev["backward_end"] = torch.cuda.Event(enable_timing=True)
            self._ev["backward_end"].record()
        for b in self._buckets:
            if b.work is None and not b.skip:
                self._launch(b)
        for b in self._buckets:
            if b.work is not None:
                b.work.wait()
            for p, v, r in zip(b.params, b.views or [], b.ready):
                if r and p.grad is not None:
                    p.grad = v
        if on_gpu:                                    # the compute stream now waits for the last collective
            self._ev["reduced"] = torch.cuda.Event(enable_timing=True)
            self._ev["reduced"].record()
        if not self._known:
            used = [bool(r and p.grad is not None) for b in self._buckets for p, r in zip(b.params, b.ready)]
            if self.world > 1 or self._reduce_at_1:
                # the grad-less set decides how many all-reduces a step issues: it must be the same on every rank, or the job
                # hangs in the collective with no diagnostic.  Ranks exchange their sets once (MAX = union, MIN =
                # intersection); where they differ the union is taken - every rank then launches the same buckets, a rank
                # whose parameter got no gradient contributes zeros - and the mismatch is counted for the caller to see
                dev = self._buckets[0].params[0].device
                m = torch.tensor([used, [not u for u in used]], dtype=torch.int32, device=dev)
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
                union, any_unused = m[0].bool().tolist(), m[1].bool().tolist()
                self.stats["used_mismatch"] = sum(1 for u, n in zip(union, any_unused) if u and n)
                used = union
            i = 0
            for b in self._buckets:
                b.used = used[i:i + len(b.params)]
                i += len(b.params)
                b.skip = not any(b.used)
                if b.views is not None:               # slots of the (now static) grad-less set stay zero from here on
                    for v, u in zip(b.views, b.used):
                        if not u:
                            v.zero_()
            self._known = True
            self.stats["skipped"] = sum(1 for b in self._buckets if b.skip)
        self.stats["steps"] += 1
        self.stats["hook_host_ms"] = 1e3 * self._hook_s
        self._armed = False
