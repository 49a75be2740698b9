"""CPU, world_size 2 (gloo): the multi-GPU path is a pure partition -- each rank owns interleaved row
bands, nothing is exchanged on the data path, and the bands concatenate to the single-device frame.
The tracer stand-in here is the CPU oracle (this is a test, there is no GPU in this container); the
GPU version of the same statement is tests/test_gpu_parity.py::test_row_stripes_equal_full_frame."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _rank_main(rank, world, port, interleaved, out_dir):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from raytracedshadows_amd import partition
    g = np.load(os.path.join(HERE, "golden", "cornell_128.npz"))
    H, W = g["mask_point"].shape
    lt = oracle.make_light(1, g["light_point"])
    mine = np.zeros((H, W), np.uint8)
    rows = partition.stripe_rows(H, world, rank, band=16, interleaved=interleaved)
    for b, e in rows:
        oracle.shadow_mask(g["packed"], g["constants"], lt, g["positions"], W, H, b, e, threads=1, out=mine)
    owned = np.zeros(H, np.int32)
    for b, e in rows:
        owned[b:e] = 1
    # control plane only: verify the partition and assemble the frame on rank 0
    t_owned = torch.from_numpy(owned)
    dist.all_reduce(t_owned)
    assert (t_owned.numpy() == 1).all()
    t = torch.from_numpy(mine.astype(np.int32))
    dist.reduce(t, dst=0)
    if rank == 0:
        assert (t.numpy().astype(np.uint8) == g["mask_point"]).all()
        open(os.path.join(out_dir, f"ok_{int(interleaved)}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_stripes_reassemble_the_frame(tmp_path):
    for i, interleaved in enumerate((True, False)):
        mp.spawn(_rank_main, args=(2, 29611 + i, interleaved, str(tmp_path)), nprocs=2, join=True)
        assert os.path.exists(os.path.join(str(tmp_path), f"ok_{int(interleaved)}"))
