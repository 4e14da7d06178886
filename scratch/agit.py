import torch, torch.distributed as dist, torch.multiprocessing as mp, os
def w(r):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("gloo", rank=r, world_size=2)
    x=torch.full((2,3),float(r)); buf=torch.empty(4,3)
    dist.all_gather_into_tensor(buf,x)
    print(r, buf.flatten().tolist())
    dist.destroy_process_group()
if __name__=="__main__":
    mp.spawn(w,nprocs=2)
