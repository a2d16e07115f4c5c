import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pioran_jl_amd as pj
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.randn(4096, dtype=torch.float64, device=dev)
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pj.farm.gather_logl(x, 4096)
    torch.cuda.synchronize(); print(i, "gather_logl ms", round(1e3 * (time.perf_counter() - t0), 3))
y = torch.empty(4096, dtype=torch.float64, device=dev)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dist.all_gather_into_tensor(y, x)
    torch.cuda.synchronize(); print(i, "raw all_gather ms", round(1e3 * (time.perf_counter() - t0), 3))
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize()
    print(i, "barrier ms", round(1e3 * (time.perf_counter() - t0), 3))
dist.destroy_process_group()
