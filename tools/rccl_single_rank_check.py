"""Development check on a one-GPU box: the RCCL calls of the N > 1 path (`init_process_group` with a
device id, all-reduce SUM on float32 / MAX on float64, barrier) with a single rank."""
import os, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda',0))
t=torch.arange(8, dtype=torch.float32, device='cuda')
dist.all_reduce(t, op=dist.ReduceOp.SUM); t.mul_(1.0)
m=torch.tensor([1.5], dtype=torch.float64, device='cuda'); dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
print('rccl ok', t.tolist(), m.item(), dist.get_backend())
dist.destroy_process_group()
