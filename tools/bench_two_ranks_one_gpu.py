"""Dry run of bench.py's N > 1 path on a ONE-GPU box: WORLD_SIZE ranks share GPU 0, the process group is
gloo and torch.distributed.gather is replaced by a host-staged gather (RCCL refuses two ranks on one
device).  Everything else -- strips, StripPipeline streams and events, the HIP kernels, the timing
protocol, the bit-for-bit check against the single-GPU frame -- is the real code.

    python tools/bench_two_ranks_one_gpu.py [world] [bench.py arguments...]
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, argv):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    real_init, real_gather = dist.init_process_group, dist.gather

    def init(backend=None, **kw):
        kw.pop("device_id", None)
        return real_init("gloo", **kw)

    def gather(tensor, gather_list=None, dst=0, group=None, async_op=False):
        torch.cuda.current_stream().synchronize()           # the strip is complete
        host = tensor.cpu()
        out = [torch.empty_like(host) for _ in gather_list] if gather_list is not None else None
        real_gather(host, out, dst=dst, group=group)
        if gather_list is not None:
            for d, s in zip(gather_list, out):
                d.copy_(s)

    dist.init_process_group, dist.gather = init, gather
    sys.argv = ["bench.py", "--gpus", str(world)] + argv
    import bench

    bench.main()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, sys.argv[2:]), nprocs=world, join=True)
