import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29655")
os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for dt in (torch.uint8, torch.float64, torch.float32):
    a=torch.arange(24, device=dev).to(dt).reshape(2,3,4); out=torch.empty_like(a)
    dist.all_gather_into_tensor(out, a); assert torch.equal(out,a)
dist.barrier(device_ids=[0]); torch.cuda.synchronize()
t=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("rccl 1-rank ok", t.item())
dist.destroy_process_group()
