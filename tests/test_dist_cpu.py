"""world_size-2 `gloo` tests of the row-split + all-gather plumbing (ggmlsharp_amd/dist.py), on CPU.

The product has no CPU compute path, so the per-shard kernel and the re-layout kernel are replaced here by checker
implementations (the oracle's mul_mat and a numpy re-layout) injected through RowSplitMulMat's hooks: what is under
test is the partition (Ggml.cs:6665-6672 over ranks), the shard padding, the collective and the [G][N][Ms] -> [N][M]
layout logic, which are backend-independent."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from ggmlsharp_amd import dist as gdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_is_the_reference_thread_split():
    # dr = ceil(nr / nth); rows [dr*ith, min(dr*(ith+1), nr))   (Ggml.cs:6665-6672)
    for M in (1, 7, 37, 4096, 32000):
        for G in (1, 2, 3, 8):
            covered = []
            for r in range(G):
                r0, r1 = gdist.shard_rows(M, G, r)
                dr = -(-M // G)
                assert r0 == min(dr * r, M) and r1 == min(dr * r + dr, M)
                covered += list(range(r0, r1))
            assert covered == list(range(M))
    assert gdist.shard_width(32000, 8) == 4000 and gdist.shard_width(37, 2) == 19


class _FakeWeight:
    def __init__(self, t, rows, K, M):
        self.type, self.rows, self.K, self.M = t, rows, K, M


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, K, N, chunks, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)          # same data on every rank
        w = rng.standard_normal((M, K)).astype(np.float32)
        x = rng.standard_normal((N, K)).astype(np.float32)
        wq = O.quantize_row(O.Q4_0, w)
        r0, r1 = gdist.shard_rows(M, world, rank)
        shard = _FakeWeight(O.Q4_0, wq[r0:r1], K, r1 - r0)

        def compute_shard(xc, out):             # checker stands in for the HIP kernel
            if shard.M == 0:
                return
            ref = O.mul_mat(O.Q4_0, shard.rows, xc.numpy(), shard.M, K, xc.shape[0])[0, 0]
            out.copy_(torch.from_numpy(ref))

        def relayout(g, G, n, Ms, Mtot, out):   # numpy statement of the re-layout kernel's index map
            gn = g.numpy().reshape(G, n, Ms)
            full = np.concatenate([gn[r] for r in range(G)], axis=1)[:, :Mtot]
            out.copy_(torch.from_numpy(np.ascontiguousarray(full)))

        def all_gather(out_t, in_t, async_op=False):
            parts = list(out_t.view(world, -1).unbind(0))
            return dist.all_gather(parts, in_t, async_op=async_op)

        runner = gdist.RowSplitMulMat(shard, N, world, rank, M_total=M, chunks=chunks, device="cpu",
                                      compute_shard=compute_shard, relayout=relayout, all_gather=all_gather)
        got = runner.step(torch.from_numpy(x)).numpy()
        ref = O.mul_mat(O.Q4_0, wq, x, M, K, N)[0, 0]
        out_q.put((rank, bool(np.array_equal(got, ref)), got.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,K,N,chunks", [(37, 64, 5, 1), (37, 64, 5, 2), (64, 128, 1, 1), (3, 64, 4, 3)])
def test_row_split_all_gather_world2(M, K, N, chunks):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, K, N, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in results:
        assert ok, f"rank {rank} result differs from the unsplit oracle"
        assert shape == (N, M)


# ---------------------------------------------------------------- two ranks on ONE GPU: the real HIP shard kernels and the
# real re-layout kernel, gloo standing in for RCCL (the driver runs the RCCL form on a whole node at round end)
def _gpu_worker(rank, world, port, M, K, N, chunks, exchange, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ggmlsharp_amd import device
        torch.cuda.set_device(0)
        device.init(0)
        rng = np.random.default_rng(9)          # same data on every rank
        w = rng.standard_normal((M, K)).astype(np.float32)
        x = rng.standard_normal((N, K)).astype(np.float32)
        wq = O.quantize_row(O.Q4_0, w)
        r0, r1 = gdist.shard_rows(M, world, rank)
        rows_dev = torch.from_numpy(wq).cuda()
        shard = device.Weight.from_device(O.Q4_0, rows_dev, K, row_begin=r0, row_end=r1)
        runner = gdist.RowSplitMulMat(shard, N, world, rank, M_total=M, chunks=chunks, exchange=exchange)   # HIP compute + HIP re-layout / peer stores
        xd = torch.from_numpy(x).cuda()
        got = runner.step(xd).clone()
        got2 = runner.step(xd).clone()          # second step reuses the buffers (the push form alternates two)
        got3 = runner.step(xd).clone()
        # the unsplit matrix on this GPU, over the same column chunks (the kernel form is a function of a call's N and K,
        # never of M: a row shard is bitwise a column slice of the unsplit result of the same call)
        Wfull = device.Weight.from_device(O.Q4_0, rows_dev, K)
        full = torch.cat([device.mul_mat(Wfull, xd[a:b]) for (a, b) in runner._chunk_bounds()], dim=0)
        ref = O.mul_mat(O.Q4_0, wq, x, M, K, N)[0, 0]
        try:                                    # THE mul_mat tolerance (tests/oracle_lib.py)
            O.assert_mul_mat_close(got.cpu().numpy(), ref, K, f"rank {rank}")
            close = True
        except AssertionError:
            close = False
        out_q.put((rank, bool(torch.equal(got, full)) and bool(torch.equal(got2, full)) and bool(torch.equal(got3, full)), close, tuple(got.shape)))
        torch.cuda.synchronize()
        dist.barrier()
        runner.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "push", "push_fused"])
@pytest.mark.parametrize("M,K,N,chunks", [(300, 256, 70, 1), (515, 512, 130, 4), (1000, 128, 1, 1), (700, 2048, 300, 2)])
def test_row_split_world2_on_one_gpu_is_bitwise_the_unsplit_result(M, K, N, chunks, exchange):
    """exchange "rccl": all-gather (gloo standing in for RCCL) + the re-layout kernel; "push": IPC-shared dst buffers across
    the two PROCESSES and the peer-store kernel (ggml_hip_ipc_*, ggml_hip_push_columns_dev)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, M, K, N, chunks, exchange, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same, close, shape in results:
        assert shape == (N, M)
        assert close, f"rank {rank}: gathered result is off the oracle"
        assert same, f"rank {rank}: gathered result is not bitwise the unsplit HIP result"
