"""Probe: can two RCCL ranks share ONE GPU on this box?  (The GPU boxes of this build have one MI355X each, so the RCCL transport of
bench.py --gpus N has never executed; if RCCL takes two ranks on one device, the multi-process control flow -- rendezvous, dmabuf IPC,
all_gather_into_tensor, d3dp_jpma_gathered on the result -- can at least run on hardware.)  Prints one JSON line."""
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, q):
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    out = {"rank": rank}
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60),
                                device_id=torch.device("cuda", 0))
        x = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(x)
        torch.cuda.synchronize()
        out["all_reduce"] = x[0].item()
        src = torch.full((1 << 22,), float(rank), device="cuda")
        dst = torch.empty((world << 22,), device="cuda")
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
        out["all_gather_ms"] = (time.perf_counter() - t0) * 1e3
        out["all_gather_ok"] = bool(all((dst[r << 22] == r).item() and (dst[((r + 1) << 22) - 1] == r).item() for r in range(world)))
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {str(e)[:600]}"
    q.put(out)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, 29631, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = []
    t_end = time.time() + 150
    while len(res) < world and time.time() < t_end:
        try:
            res.append(q.get(timeout=5))
        except Exception:           # noqa: BLE001
            if not any(p.is_alive() for p in ps):
                break
    for p in ps:
        p.join(timeout=5)
        if p.is_alive():
            p.kill()
    print(json.dumps({"world": world, "devices": torch.cuda.device_count(), "results": sorted(res, key=lambda r: r["rank"])}))
