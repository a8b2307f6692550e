"""The collective path of the row-sharded matrix EXECUTED on hardware (round 5): a 1-rank "nccl" (= RCCL) process group
on the one GPU of the box, ShardedMatrix.step() forced through dist.all_gather_into_tensor.  World 1 moves no bytes
between GPUs, but it runs everything else an 8-rank node runs: RCCL initialisation, the collective enqueued by torch
behind the engine's row solve on the backend's stream (EngineBackend.stream: the fork / join of the bin launches must be
ordered in front of it), the int64 tensor views of the bit rows, and the assemble + set-max behind it.  The multi-rank
layout itself is covered by tests/test_dist_gloo.py (world 2 / 3 on CPU) and the one-GPU shard emulations of
tests/test_gpu_full_configs.py; the 8-GPU scaling curve is the driver's."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group exists already")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("workload", ["tiny", "C1"])
def test_step_through_the_rccl_all_gather_equals_the_plain_step(rccl_group, workload):
    import bench
    import torch
    from ipc_amd.consensus import IPC
    from ipc_amd.dist import EngineBackend, ShardedMatrix
    g, cfg, _ = bench.build_workload(workload)
    eng = IPC(g, cfg, device=0)
    plain = ShardedMatrix(EngineBackend(eng), 0, 1)
    plain.step()
    bits0, acc0 = plain.result()
    coll = ShardedMatrix(EngineBackend(eng), 0, 1, force_gather=True)
    assert coll.gathered.data_ptr() != coll.upper.data_ptr()
    for _ in range(3):                                  # back to back: the next row solve may not overtake the collective
        coll.gathered.fill_(-1)
        coll.step()
    bits1, acc1 = coll.result()
    torch.cuda.synchronize()
    assert np.array_equal(bits0, bits1)
    assert np.array_equal(acc0, acc1)
    assert int(acc1.sum()) > 0
    eng.close()


def test_all_gather_is_ordered_behind_the_row_solve_on_the_backend_stream(rccl_group):
    """The gathered rows are exactly the rank's shard of THIS step: the shard buffer is poisoned before every step, so a
    collective that ran ahead of the row solve (unordered streams) would gather the poison."""
    import bench
    import torch
    from ipc_amd.consensus import IPC
    from ipc_amd.dist import EngineBackend, ShardedMatrix
    g, cfg, _ = bench.build_workload("T700")
    eng = IPC(g, cfg, device=0)
    sm = ShardedMatrix(EngineBackend(eng), 0, 1, force_gather=True)
    sm.step()
    ref = sm.gathered.clone()
    torch.cuda.synchronize()
    for _ in range(4):
        with sm.b.stream_ctx():
            sm.upper.fill_(0x5a5a5a5a5a5a5a5a)
        sm.step()
        sm.b.stream.synchronize()
        assert torch.equal(sm.gathered, ref)
    eng.close()
