"""RCCL smoke (one rank): high-priority ProcessGroupNCCL options + an all-gather on a side stream, as bench.py --gpus N uses them."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0), pg_options=opts)
s = torch.cuda.Stream()
x = torch.arange(4, dtype=torch.int64, device="cuda"); g = torch.zeros((1,4), dtype=torch.int64, device="cuda")
ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); s.wait_event(ev)
with torch.cuda.stream(s):
    dist.all_gather_into_tensor(g.view(-1), x)
dist.barrier(); torch.cuda.synchronize()
print("ok", g.tolist())
dist.destroy_process_group()
