"""Row-split of the weight matrix across the GPUs of one node + exchange of the dst shards (one process per GPU).

The reference's only parallelism is a contiguous row partition of src0 over CPU threads
(Ggml.cs:6665-6672: dr = ceil(nr / nth), thread ith owns rows [dr*ith, min(dr*(ith+1), nr))).  The same partition
is used over ranks: one process per GPU, rank r keeps its row shard resident, every rank holds all of src1 and quantizes
it locally (deterministic, identical on every rank), and the dst shards are exchanged so that every rank ends with the
reference's dst layout [N][M] (Ggml.cs:6692-6697).  Two exchange forms, identical bits (they only move data):

  "rccl"  dst shards [N][Ms] (m fastest) -> one all_gather_into_tensor per src1-row chunk (RCCL over xGMI under the
          "nccl" backend), issued async so chunk i's gather overlaps chunk i+1's kernels -> the re-layout kernel
          [G][N][Ms] -> [N][M] (SURVEY.md 8(e) layout catch; skipped for N == 1, where the gathered buffer IS the layout).
  "push"  every rank's [N][M] result buffer is IPC-shared; a rank computes its rows straight into its own buffer's
          columns and one kernel stores them into every peer's buffer (compute units over xGMI: one hop, all links of the
          sender at once, final layout -- no [G][N][Ms] intermediate, no re-layout pass); one tiny all-reduce per step is
          the barrier.  Two result buffers alternate, so a step's stores never race the previous step's consumers.
  "push_fused" (r4)  the same buffers, but the stores to the peers are the GEMM's OWN store phase (ggml_hip_mul_mat_push_dev:
          mm_epilogue mode 3, up to eight destination bases) where the kernel form has it -- the staged MX forms and K3p --
          so a step is INIT + one COMPUTE launch whose tiles go out over xGMI as they finish; forms without it fall back to
          "push" for that call.  SURVEY 8(e): "epilogue peer-writes straight into each peer's final [N][M] buffer".
"""
import ctypes as C

import torch


def shard_rows(M, world, rank):
    """The reference's thread split (Ggml.cs:6665-6672) over ranks."""
    dr = (M + world - 1) // world
    r0 = min(dr * rank, M)
    r1 = min(r0 + dr, M)
    return r0, r1


def shard_width(M, world):
    return (M + world - 1) // world


class _RawDeviceBuffer:
    """A raw device allocation (ggml_hip_ipc_alloc / _open) seen by torch without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr, shape):
        self.ptr = ptr
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class RowSplitMulMat:
    """dst = mul_mat(W, x) with W row-split over `world` ranks.

    compute_shard(x_chunk, out_chunk) and relayout(gathered, G, n, Ms, M, out) default to the HIP path; tests of the
    distributed plumbing on CPU (gloo) inject checker implementations instead -- there is no CPU path in the product.
    """

    def __init__(self, weight, N, world, rank, M_total=None, chunks=1, device=None, compute_shard=None, relayout=None,
                 all_gather=None, exchange="rccl", group_backend=None):
        self.W, self.N, self.world, self.rank = weight, N, world, rank
        self.Ms = weight.M if world == 1 else None
        if world > 1:
            self.M_total = M_total if M_total is not None else weight.M * world
            self.Ms = shard_width(self.M_total, world)
        else:
            self.M_total = weight.M
        self.chunks = max(1, min(chunks, N)) if world > 1 else 1
        self.fused_store = exchange == "push_fused"
        if exchange == "push_fused":
            exchange = "push"
        self.exchange = exchange if world > 1 else "none"
        self._default_compute = compute_shard is None
        dev = device if device is not None else "cuda"
        self.dev = dev
        self.work = None
        self._push = None
        if compute_shard is None:
            from . import device as D
            self.work = D.alloc_work(weight.type, weight.K, N, dev)

            def compute_shard(x, out, _D=D, _w=weight, _work=self.work):
                _D.mul_mat(_w, x, out=out, work=_work)

            def relayout_default(g, G, n, Ms, M, out, _D=D):
                _D.relayout_gathered(g, G, n, Ms, M, out=out)
            relayout = relayout or relayout_default
        self.compute_shard = compute_shard
        self.relayout = relayout
        if self.exchange == "push":
            self._setup_push(group_backend)
            return
        self.shard = torch.zeros((N, self.Ms), dtype=torch.float32, device=dev)  # zero: pad columns of a short last shard
        if world > 1:
            import torch.distributed as dist
            self.all_gather = all_gather or dist.all_gather_into_tensor
            self.gathered = torch.empty((world * N * self.Ms,), dtype=torch.float32, device=dev)
            self.out = torch.empty((N, self.M_total), dtype=torch.float32, device=dev)

    # ---------------------------------------------------------------- "push": IPC-shared result buffers
    def _setup_push(self, group_backend):
        """Every rank runs the SAME sequence of collectives whatever fails locally (a rank that raised early while its peers
        sat in a collective would leave the job out of step): handles are exchanged, every rank tries to open its peers',
        one all-reduce tells everybody whether EVERY rank succeeded, and only then does anyone raise."""
        import torch.distributed as dist
        from ._lib import check, lib
        L = lib()
        self._L = L
        nbytes = self.N * self.M_total * 4
        self._own, self._peers, self._views, self._opened = [], [], [], []
        handles = torch.zeros((2, 64), dtype=torch.uint8)
        err = None
        try:
            for b in range(2):
                p = C.c_void_p()
                h = (C.c_uint8 * 64)()
                check(L.ggml_hip_ipc_alloc(nbytes, C.byref(p), h), "ggml_hip_ipc_alloc")
                self._own.append(p.value)
                handles[b] = torch.tensor(list(h), dtype=torch.uint8)
        except Exception as e:  # noqa: BLE001 -- reported after the collectives below
            err = e
        on_gpu = (group_backend or dist.get_backend()) == "nccl"
        mine = handles.cuda() if on_gpu else handles
        allh = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allh, mine)
        # which device every rank computes on: a store into a peer's buffer needs peer access between the two devices, and a
        # store without it is a GPU fault, not an error code -- ask before any kernel touches a peer pointer
        di = torch.device(self.dev).index
        my_dev = di if di is not None else torch.cuda.current_device()
        devs = torch.tensor([my_dev], dtype=torch.int64)
        devs = devs.cuda() if on_gpu else devs
        alld = [torch.empty_like(devs) for _ in range(self.world)]
        dist.all_gather(alld, devs)
        if err is None:
            for r in range(self.world):
                pd = int(alld[r].item())
                if r != self.rank and pd != my_dev and not torch.cuda.can_device_access_peer(my_dev, pd):
                    err = RuntimeError(f"device {my_dev} has no peer access to device {pd} (rank {r})")
                    break
        if err is None:
            try:
                for b in range(2):
                    ptrs = []
                    for r in range(self.world):
                        if r == self.rank:
                            ptrs.append(self._own[b])
                            continue
                        hb = (C.c_uint8 * 64)(*allh[r][b].cpu().tolist())
                        p = C.c_void_p()
                        check(L.ggml_hip_ipc_open(hb, C.byref(p)), "ggml_hip_ipc_open")
                        self._opened.append(p.value)
                        ptrs.append(p.value)
                    self._peers.append(ptrs)
                    self._views.append(torch.as_tensor(_RawDeviceBuffer(self._own[b], (self.N, self.M_total)), device=self.dev))
            except Exception as e:  # noqa: BLE001
                err = e
        ok = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float32, device=self.dev if on_gpu else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            self._views = []
            for p in self._opened:
                L.ggml_hip_ipc_close(C.c_void_p(p))
            self._opened = []
            dist.barrier()                                 # nobody frees a buffer a peer still has mapped
            for p in self._own:
                L.ggml_hip_ipc_free(C.c_void_p(p))
            self._own = None
            raise RuntimeError(f"push exchange unavailable on at least one rank (this rank: {err!r})")
        self._flag = torch.zeros(1, dtype=torch.float32, device=self.dev if on_gpu else "cpu")
        self._on_gpu = on_gpu
        self._side = torch.cuda.Stream()
        self._turn = 0
        self.out = self._views[0]
        self.shard = None
        dist.barrier()

    def close(self):
        """Release the IPC mappings and the shared buffers of the "push" form (after a barrier of the caller's)."""
        if self.exchange != "push" or self._own is None:
            return
        for p in self._opened:
            self._L.ggml_hip_ipc_close(C.c_void_p(p))
        self._views = []
        self.out = None
        for p in self._own:
            self._L.ggml_hip_ipc_free(C.c_void_p(p))
        self._own = None

    def _chunk_bounds(self):
        step = (self.N + self.chunks - 1) // self.chunks
        return [(a, min(a + step, self.N)) for a in range(0, self.N, step)]

    def _step_push(self, x):
        import torch.distributed as dist
        from ._lib import check
        b = self._turn
        self._turn ^= 1
        out = self._views[b]
        r0, r1 = shard_rows(self.M_total, self.world, self.rank)
        Mw = r1 - r0
        main = torch.cuda.current_stream()
        side = self._side                                               # the stores of chunk i overlap the kernels of chunk i + 1
        peers = (C.c_void_p * self.world)(*[None if r == self.rank else p for r, p in enumerate(self._peers[b])])
        for (a, e) in self._chunk_bounds():
            if Mw > 0 and self.fused_store and self._default_compute and self._L.ggml_hip_mul_mat_push_fused(self.W.handle, e - a, self.world):
                # the exchange IS the GEMM's store phase: every destination's base at this chunk's first row, this rank's own among them
                xa = x[a:e]
                allp = (C.c_void_p * self.world)(*[p + a * self.M_total * 4 for p in self._peers[b]])
                check(self._L.ggml_hip_mul_mat_push_dev(self.W.handle, C.c_void_p(xa.data_ptr()), e - a, xa.stride(0), allp, self.world, self.rank,
                                                        self.M_total, r0, C.c_void_p(self.work.data_ptr()), self.work.numel(),
                                                        C.c_void_p(main.cuda_stream)), "ggml_hip_mul_mat_push_dev")
                continue
            if Mw > 0:
                own = out[a:e, r0:r1]                                   # this rank's rows as columns of its own dst
                self.compute_shard(x[a:e], own)
                side.wait_stream(main)
                src = self._own[b] + (a * self.M_total + r0) * 4
                shifted = (C.c_void_p * self.world)(*[None if p is None else p + a * self.M_total * 4 for p in peers])
                check(self._L.ggml_hip_push_columns_dev(C.c_void_p(src), self.M_total, e - a, Mw, shifted, self.world,
                                                        self.M_total, r0, C.c_void_p(side.cuda_stream)), "ggml_hip_push_columns_dev")
        main.wait_stream(side)
        # barrier: every rank's stores of this step have been issued and completed before anyone reads `out`
        if self._on_gpu:
            dist.all_reduce(self._flag)
        else:
            torch.cuda.synchronize()
            dist.barrier()
        self.out = out
        return out

    def step(self, x):
        if self.world == 1:
            self.compute_shard(x, self.shard)
            return self.shard
        if self.exchange == "push":
            return self._step_push(x)
        Mw = self.W.M  # rows this rank really owns (the last rank may own fewer than Ms)
        pending = []
        off = 0
        for (a, b) in self._chunk_bounds():
            n = b - a
            self.compute_shard(x[a:b], self.shard[a:b, :Mw])
            g = self.gathered[off: off + self.world * n * self.Ms].view(self.world, n, self.Ms)
            off += self.world * n * self.Ms
            h = self.all_gather(g.view(-1), self.shard[a:b].reshape(-1), async_op=True)
            pending.append((h, g, a, b))
        for (h, g, a, b) in pending:
            if h is not None:
                h.wait()
            if self.N == 1:
                self.out.view(-1)[: self.M_total].copy_(g.view(-1)[: self.M_total])
            else:
                self.relayout(g, self.world, b - a, self.Ms, self.M_total, self.out[a:b])
        return self.out

    # ---- timing probes for bench.py: the phases alone ----
    def compute_only(self, x):
        """INIT + COMPUTE of this rank's shard for every chunk, no exchange."""
        if self.exchange == "push":
            r0, r1 = shard_rows(self.M_total, self.world, self.rank)
            for (a, e) in self._chunk_bounds():
                if r1 > r0:
                    self.compute_shard(x[a:e], self._views[0][a:e, r0:r1])
            return
        Mw = self.W.M
        for (a, b) in self._chunk_bounds():
            self.compute_shard(x[a:b], self.shard[a:b, :Mw])

    def exchange_only(self):
        """The exchange of already computed shards (+ re-layout / barrier), no kernels of the hot path."""
        import torch.distributed as dist
        if self.world == 1:
            return
        if self.exchange == "push":
            from ._lib import check
            r0, r1 = shard_rows(self.M_total, self.world, self.rank)
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            peers = (C.c_void_p * self.world)(*[None if r == self.rank else p for r, p in enumerate(self._peers[0])])
            if r1 > r0:
                check(self._L.ggml_hip_push_columns_dev(C.c_void_p(self._own[0] + r0 * 4), self.M_total, self.N, r1 - r0, peers,
                                                        self.world, self.M_total, r0, stream), "ggml_hip_push_columns_dev")
            if self._on_gpu:
                dist.all_reduce(self._flag)
            else:
                torch.cuda.synchronize()
                dist.barrier()
            return
        h = self.all_gather(self.gathered, self.shard.reshape(-1), async_op=True)
        if h is not None:
            h.wait()
        if self.N > 1:
            self.relayout(self.gathered.view(self.world, self.N, self.Ms), self.world, self.N, self.Ms, self.M_total, self.out)
