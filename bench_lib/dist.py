"""bench.py's process group: one rank per GPU over RCCL (or gloo), and the library's own communicator for the exchanges."""
import os


def init_distributed(args, world, local_rank):
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            from xworld_amd.sharding import init_nccl
            init_nccl(torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend)
        return local_rank
    torch.cuda.set_device(0)
    if args.force_exchange:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from xworld_amd.sharding import nccl_init_kwargs
        kw = nccl_init_kwargs(torch.device("cuda", 0)) if args.backend == "nccl" else {}
        if "MASTER_ADDR" in os.environ and "MASTER_PORT" in os.environ:
            dist.init_process_group(args.backend, world_size=1, rank=0, **kw)
        else:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, **kw)
    return local_rank


def library_comm(args, world, rank, local_rank, dev):
    """--exchange lib | auto: the library's own communicator, made in a thread with a deadline: a second ncclCommInitRank that
    hangs on some box must cost the faster exchange, not the run; every rank then agrees (over the torch group) on what to use"""
    import threading
    import torch
    import torch.distributed as dist
    from xworld_amd import sharding
    box = {}

    def make_comm():
        try:
            box["comm"] = sharding.LibComm(rank, world, local_rank)
        except Exception as e:                           # noqa: BLE001
            box["err"] = "%s: %s" % (type(e).__name__, e)
    th = threading.Thread(target=make_comm, daemon=True)
    th.start()
    th.join(90.0)
    ok = torch.tensor([1 if "comm" in box else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()):
        return box["comm"], None
    note = box.get("err", "timed out after 90 s" if th.is_alive() else "another rank failed")
    if args.exchange == "lib":
        raise RuntimeError("--exchange lib: the library's communicator did not come up: %s" % note)
    return None, note
