"""Software pipeline of query batches: slots that overlap consecutive batches on the device.

A batch is five launches (r06): `encode_hash` with the bucket lookup of the scan's PLAN phase in its epilogue, two small latency-bound
kernels that lay out the (row segment, query group) tasks, the scan kernel, which fills the chip, and the per-query merge of the partial
lists -- followed, on a sharded index, by the all-gather + shard merge.  Run back to back on one stream, everything but the scan costs
~75 us per batch during which most of the chip idles (and the collective's latency is exposed).  Two kinds of slot overlap them:

* graph slots (r06, the default for the bucket-major schedules; `nlsh_step_create_graph`, ABI v4): a slot owns a stream, the batch's five
  launches are captured ONCE into a hipGraph, and `submit` replays it on that stream after the library has refreshed the two kernel
  nodes that carry the batch pointer.  Batches of different slots overlap because their streams do; consecutive batches of one slot
  are ordered by its stream.  Host cost per batch: one ctypes transition = one graph launch + two node updates + one event record
  (0.02 ms; the staged slots' five launches + eight to ten event calls cost 0.05-0.10).  Two scan kernels may share the chip, so a scan
  bracketed by events is not timed alone (a batch submitted WITH scan events is launched eagerly on the slot's stream instead);
* staged slots (r03-r05, `graph=False`): the stages of consecutive batches on three (four) shared streams

      front:  encode(i+1) plan(i+1)           | encode(i+2) plan(i+2) | ...
      mid  :  scan(i)                         | scan(i+1)             | ...
      tail :  merge(i-1) [all-gather(i-1)]    | merge(i) ...          | ...

  where scan kernels never overlap each other (their HIP-event durations stay meaningful: +7 %).

`depth` slots own the per-batch buffers (key table, task-table workspace, outputs); a slot is reused only after its previous batch has
finished (its own stream's order, or the tail event).  Results are bit-identical to `Indexer.query_tensors`: the kernels and their
arguments are the same, only the stream they run on differs.  No reference counterpart (the reference answers one query at a time,
nlsh/indexer.py:62-95).

Buffer lifetimes: a submitted batch is read by all stages after `submit` returns.  The pipeline keeps a reference to the batch tensor
in the batch's slot, so the caller may drop it at once, but must not OVERWRITE it in place before `synchronize()` (or the batch's
results) say the batch is done.  When the slot is reused while its previous batch is still in flight (the host running a whole
pipeline depth ahead of the device), the old tensor is first marked as in use on the stage streams (`record_stream`), so the caching
allocator will not hand its memory out until those streams have passed it.  The packed encoder weights are owned by the pipeline (a
reference is held) and re-read from the hasher whenever its parameters changed since the last submit (a training step, `load_state`,
a device move), so slots never keep a dangling pointer.  A submit that raises leaves its slot's workspace in an undefined state (the
PLAN phase keeps counters at its head zero between calls, include/nlsh_hip.h): build a new pipeline after an error.
"""
from typing import Callable, Optional

import torch

from . import _capi


class _Slot:
    pass


class QueryPipeline:

    default_graph = True     # slots replay a captured hipGraph of the batch on a stream of their own (False: the staged streams of r03-r05)

    def __init__(self, indexer, sample_queries, k=10, hash_times=10, depth=3, want_keys=False,
                 exchange: Optional[Callable] = None, split_front: Optional[bool] = None, graph: Optional[bool] = None):
        """`sample_queries`: a batch of the shape every later batch has (sizes the task table with one ordinary,
        checked call).  `exchange(keys64, ncand) -> (dist, idx, ncand)`: the sharded index's all-gather + merge,
        run on the tail stream (needs the 64-bit keys, so it implies want_keys).
        `graph` (r06, ABI v4): True = every slot replays a captured hipGraph of the batch's five launches on a stream of its OWN
        (`nlsh_step_create_graph`): batches overlap because the slots' streams do, and a submit costs the host one graph launch + two
        node updates instead of the staged slots' five launches + eight to ten event calls.  False = the three / four stage streams of
        r03-r05.  None = graph slots whenever the schedule is a bucket-major one (`QueryPipeline.default_graph`)."""
        if hash_times > _capi.MAX_PROBES:
            raise _capi.NlshHipError(_capi.E_UNSUPPORTED, f"pipelined batches take hash_times <= {_capi.MAX_PROBES}")
        if depth < 2:
            raise ValueError("depth must be >= 2 (3 keeps all three stages busy)")
        self.indexer, self.k, self.P, self.exchange = indexer, k, hash_times, exchange
        q = sample_queries
        if q.dtype != torch.float32 or q.stride(1) != 1:
            raise ValueError("pipelined batches must be float32 row-major device tensors")
        dev = q.device
        self.Q, self.d = q.shape
        indexer.query_tensors(q, k=k, hash_times=hash_times, seed=0, check=True)     # sizes indexer._max_tasks
        torch.cuda.synchronize(dev)
        if graph is None:
            graph = self.default_graph
        self.graph = bool(graph) and indexer.last_algo != _capi.SCAN_QUERY_MAJOR and indexer.n_buckets > 0
        if self.graph:
            split_front = False
        if split_front is None:
            # Four stages (the PLAN phase on a stream of its own behind the encode) pay when the front stage -- encode + PLAN, two
            # latency-bound chains of ~40 us each whatever the shard size -- is about as long as the scan: small shards of a
            # multi-GPU run.  Measured on the one-GPU emulation of the headline's shards (r04): 1 shard 0.290 -> 0.296 ms per
            # pipelined step, 4 shards 0.128 -> 0.119, 8 shards 0.108 -> 0.095 (then equal to the host's time to enqueue a step).
            # Decided from two timed sequential steps of THIS batch shape on THIS index: the whole step against its scan kernel.
            cur = torch.cuda.current_stream(dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for e in ev:
                e.record(cur)
            for _ in range(2):
                ev[0].record(cur)
                indexer.query_tensors(q, k=k, hash_times=hash_times, seed=0, check=False, events=(ev[1], ev[2]))
                ev[3].record(cur)
            torch.cuda.synchronize(dev)
            step_ms, scan_ms = ev[0].elapsed_time(ev[3]), ev[1].elapsed_time(ev[2])
            split_front = scan_ms < 1.25 * 0.8 * max(step_ms - scan_ms, 0.0)   # ~80 % of what is not the scan is the front stage
        self.split_front = bool(split_front)
        if self.split_front and depth < 4:
            depth = 4                                                 # one slot per stage in flight
        self.algo = indexer.last_algo
        self.max_tasks = indexer._max_tasks[indexer._last_tkey]
        ws_bytes = _capi.lib().nlsh_scan_workspace(self.Q, self.P, k, self.max_tasks, indexer.n_buckets, self.d)
        # the front and tail stages are small workgroups that must slip in beside the scan: their queues get the
        # higher priority
        self.front = torch.cuda.Stream(device=dev, priority=-1)
        self.mid = torch.cuda.Stream(device=dev, priority=0)
        self.tail = torch.cuda.Stream(device=dev, priority=-1)
        # split_front: the PLAN phase on a stream of its own behind the encode (four stages): on small shards the front stage
        # (encode + PLAN, two latency-bound chains of ~40 us each) is the longest one and bounds the pipelined step
        self.plan = torch.cuda.Stream(device=dev, priority=-1) if self.split_front else None
        self.slots = []
        for _ in range(depth):
            s = _Slot()
            s.lane = torch.cuda.Stream(device=dev) if self.graph else None    # graph slots: one stream per slot
            s.keys = torch.empty((self.Q, self.P), dtype=torch.int32, device=dev)
            s.nkeys = torch.empty((self.Q,), dtype=torch.int32, device=dev)
            s.out_dist = torch.empty((self.Q, k), dtype=torch.float32, device=dev)
            s.out_idx = torch.empty((self.Q, k), dtype=torch.int32, device=dev)
            s.out_keys = torch.empty((self.Q, k), dtype=torch.int64, device=dev) if (want_keys or exchange) else None
            s.ncand = torch.empty((self.Q,), dtype=torch.int32, device=dev)
            s.status = torch.zeros((2,), dtype=torch.int32, device=dev)
            s.ws = torch.zeros((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)   # PLAN-phase head must start out zero (include/nlsh_hip.h)
            s.batch = None
            s.step = None
            self.slots.append(s)
        self._lib = _capi.lib()
        self._n_multi = indexer._n_multi_rows(self.Q)
        self._hold_done = exchange is not None
        self._packed = None
        self._bind_weights()
        self.n_submitted = 0
        self.last_slot = None
        # the slots' buffers were zero-filled (and the weights packed) on the caller's current stream; the stage streams do not order
        # themselves behind it, and the first `submit` may come from another stream: nothing of the set-up is left in flight
        torch.cuda.current_stream(dev).synchronize()

    def _make_step(self, s):
        """The slot's `nlsh_step_t`: everything about its launches that does not change from batch to batch."""
        import ctypes
        ix, h = self.indexer, self.indexer._hashing
        (n_layers, dims_arr, packed_ptr, act, key_mode, n_probes), _ = h.encode_args(self.P, s.keys, s.nkeys)
        pre, post = ix._scan_args(self.Q, self.d, s.keys, s.nkeys, self.k, self.algo, self.max_tasks, s.out_dist, s.out_idx, s.out_keys,
                                  s.ncand, s.status, s.ws)
        (corpus, row_stride, d, gid, uniq, offsets, order, n_buckets, cell_of, cell_offsets, n_cells, inv_norm) = pre
        (Q, qkeys, nkeys, P, k, metric, algo, seg, out_dist, out_idx, out_keys, ncand, status, ws, ws_bytes, max_tasks) = post
        desc = _capi.StepDesc(
            n_layers=n_layers, act=act, key_mode=key_mode, n_probes=n_probes, dims=ctypes.cast(dims_arr, ctypes.c_void_p), packed=packed_ptr,
            n_multi_rows=self._n_multi, corpus_sorted=corpus, row_stride=row_stride, gid=gid, uniq_keys=uniq, offsets=offsets,
            bucket_order=order, cell_of=cell_of, cell_offsets=cell_offsets, inv_norm=inv_norm, d=d, n_buckets=n_buckets, n_cells=n_cells,
            k=k, metric=metric, algo=algo, seg_rows=seg, hold_done=int(self._hold_done), Q=Q, qkeys=qkeys, nkeys=nkeys, out_dist=out_dist,
            out_idx=out_idx, out_keys=out_keys, out_ncand=ncand, status=status, workspace=ws, workspace_bytes=ws_bytes, max_tasks=max_tasks,
            front=self.front.cuda_stream, plan=self.plan.cuda_stream if self.plan is not None else None, mid=self.mid.cuda_stream,
            tail=self.tail.cuda_stream)
        handle = ctypes.c_void_p()
        if self.graph:
            _capi.check(self._lib.nlsh_step_create_graph(ctypes.byref(desc), ctypes.sizeof(desc), s.lane.cuda_stream, ctypes.byref(handle)))
        else:
            _capi.check(self._lib.nlsh_step_create(ctypes.byref(desc), ctypes.sizeof(desc), ctypes.byref(handle)))
        return handle

    def _bind_weights(self):
        """Point every slot at the hasher's CURRENT packed weights and keep the blob alive for as long as the slots point at it."""
        h = self.indexer._hashing
        packed = h.packed_weights()
        for st in ([sl.lane for sl in self.slots] if self.graph else [self.front]):
            packed.record_stream(st)          # read by encode_hash on the front stream (the slots' lanes), allocated on the caller's
        self._weights_sig = h._weights_signature()
        for s in self.slots:
            if s.step is None:
                self._packed = packed
                s.step = self._make_step(s)
            else:
                _capi.check(self._lib.nlsh_step_set_weights(s.step, packed.data_ptr()))
        self._packed = packed

    def close(self):
        """Destroy the slots' library objects (after the batches in flight have finished)."""
        if getattr(self, "slots", None):
            self.synchronize()
            for s in self.slots:
                if s.step is not None:
                    self._lib.nlsh_step_destroy(s.step)
                    s.step = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the process is going away with the handles
            pass

    def _stage_streams(self, slot=None):
        if self.graph:
            return (slot.lane,) if slot is not None else tuple(sl.lane for sl in self.slots)
        return (self.front, self.mid, self.tail) + ((self.plan,) if self.plan is not None else ())

    def submit(self, queries, seed=None, events=None):
        """Enqueue one batch; returns (dist, idx, ncand, keys64 | None) -- device tensors owned by the batch's slot
        (or fresh ones from `exchange`), valid once the tail stream has passed the batch (`synchronize()`), and
        overwritten `depth` submits later.  `events`: (begin, end) pair recorded around the scan kernel.
        The hasher's weights are the ones present at THIS call.  A batch that needs more tasks than the sample batch's table holds
        (1.25x + 1024 of what the sample needed) is reported by `overflowed()`, not by this call: check it before trusting a result
        of a batch much heavier than the sample."""
        if queries.shape != (self.Q, self.d) or queries.dtype != torch.float32 or queries.stride(1) != 1:
            raise ValueError("batch shape/dtype differs from the pipeline's sample batch")
        ix, L = self.indexer, self._lib
        if ix._hashing._needs_train_forward():
            # the pipeline launches the fused kernel on folded eval-mode weights; a BatchNorm encoder in train mode needs the module's
            # own batch-statistics forward (hashings._run_train_mode), which `Indexer.query` / `hash_device` route to
            raise _capi.NlshHipError(_capi.E_UNSUPPORTED, "pipelined batches need the hasher in eval mode (BatchNorm encoder in train "
                                                          "mode: call hashing.train_mode(False), or use Indexer.query)")
        s = self.slots[self.n_submitted % len(self.slots)]
        self.n_submitted += 1
        if ix._hashing._weights_signature() != self._weights_sig:
            if self.n_submitted > 1:
                for st in ([sl.lane for sl in self.slots] if self.graph else [self.front]):
                    torch.cuda.current_stream(queries.device).wait_stream(st)       # batches in flight still read the old blob
            self._bind_weights()
        if s.batch is not None and s.batch is not queries and L.nlsh_step_busy(s.step) != 0:
            # the slot's previous batch is still in flight and its tensor is about to lose our reference: keep its memory away from
            # the allocator until the stage streams have passed it (the host runs a pipeline depth ahead: the device-bound regime,
            # where these calls cost nothing that matters)
            for st in self._stage_streams(s):
                s.batch.record_stream(st)
        if seed is None:
            seed = ix._hashing.next_seed()
        rc = L.nlsh_query_step_enqueue(s.step, queries.data_ptr(), queries.stride(0), seed, torch.cuda.current_stream(queries.device).cuda_stream,
                                       events[0].cuda_event if events else None, events[1].cuda_event if events else None)
        s.batch = queries
        _capi.check(rc)
        out = (s.out_dist, s.out_idx, s.ncand, s.out_keys)
        if self.exchange is not None:
            with torch.cuda.stream(s.lane if self.graph else self.tail):
                out = tuple(self.exchange(s.out_keys, s.ncand)) + (None,)
            _capi.check(L.nlsh_step_release(s.step))
        self.last_slot = s
        return out

    def synchronize(self):
        if self.graph:
            for s in self.slots:
                s.lane.synchronize()
        else:
            self.tail.synchronize()

    def overflowed(self) -> bool:
        """True if any slot's last batch did not fit the task table (results incomplete: rebuild the pipeline).
        Raises if a slot's PLAN phase found its workspace head non-zero on entry (status 2: the batch got no tasks)."""
        flags = [int(s.status.cpu()[1]) for s in self.slots]
        if 3 in flags:
            raise _capi.NlshHipError(_capi.E_INVALID, "pipeline slot: the cells hold a shared window wider than one 256-row segment")
        if 2 in flags:
            raise _capi.NlshHipError(_capi.E_WORKSPACE, "pipeline slot: the pair counters at the head of the workspace were not "
                                                        "zero on entry (workspace contract, include/nlsh_hip.h)")
        return any(flags)
