"""Row-split of the weight matrix across the GPUs of one node + all-gather of the dst shards.

The reference's only parallelism is a contiguous row partition of src0 over CPU threads
(Ggml.cs:6665-6672: dr = ceil(nr / nth), thread ith owns rows [dr*ith, min(dr*(ith+1), nr))).  The same partition
is used over ranks: one process per GPU, rank r keeps its row shard resident, every rank holds all of src1, and
the dst shards ([N][Ms] each, m fastest) are exchanged with one all-gather (RCCL over xGMI under the "nccl"
backend) and re-laid-out into the reference's dst layout [N][M] (Ggml.cs:6692-6697; SURVEY.md 8(e) layout catch).
For N == 1 the gathered buffer already is that layout and the re-layout is skipped.

With `chunks` > 1 src1 rows are processed in chunks and each chunk's gather is issued asynchronously, so the
exchange of chunk i overlaps the kernels of chunk i+1.
"""
import torch


def shard_rows(M, world, rank):
    """The reference's thread split (Ggml.cs:6665-6672) over ranks."""
    dr = (M + world - 1) // world
    r0 = min(dr * rank, M)
    r1 = min(r0 + dr, M)
    return r0, r1


def shard_width(M, world):
    return (M + world - 1) // world


class RowSplitMulMat:
    """dst = mul_mat(W, x) with W row-split over `world` ranks.

    compute_shard(x_chunk, out_chunk) and relayout(gathered, G, n, Ms, M, out) default to the HIP path; tests of the
    distributed plumbing on CPU (gloo) inject checker implementations instead -- there is no CPU path in the product.
    """

    def __init__(self, weight, N, world, rank, M_total=None, chunks=1, device=None, compute_shard=None, relayout=None,
                 all_gather=None):
        self.W, self.N, self.world, self.rank = weight, N, world, rank
        self.Ms = weight.M if world == 1 else None
        if world > 1:
            self.M_total = M_total if M_total is not None else weight.M * world
            self.Ms = shard_width(self.M_total, world)
        else:
            self.M_total = weight.M
        self.chunks = max(1, min(chunks, N)) if world > 1 else 1
        dev = device if device is not None else "cuda"
        self.shard = torch.zeros((N, self.Ms), dtype=torch.float32, device=dev)  # zero: pad columns of a short last shard
        self.work = None
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
        if world > 1:
            import torch.distributed as dist
            self.all_gather = all_gather or dist.all_gather_into_tensor
            self.gathered = torch.empty((world * N * self.Ms,), dtype=torch.float32, device=dev)
            self.out = torch.empty((N, self.M_total), dtype=torch.float32, device=dev)

    def _chunk_bounds(self):
        step = (self.N + self.chunks - 1) // self.chunks
        return [(a, min(a + step, self.N)) for a in range(0, self.N, step)]

    def step(self, x):
        if self.world == 1:
            self.compute_shard(x, self.shard)
            return self.shard
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
