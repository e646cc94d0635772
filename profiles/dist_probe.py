import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533"); os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda",0))
from pixelbox_amd import capi, synth
n=1_250_000
ix=capi.Index(256,n); ix.fill_synthetic(synth.SEED_INDEX,0,n,1); ix.set_option(capi.PB_OPT_SEARCH_PATH,2)
q=synth.fill_synthetic(synth.SEED_QUERY,0,64*256).reshape(64,256)
k=100
dev=torch.device("cuda",0)
def T(f, reps=20):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e3
packed=torch.empty((64,2*k+1),dtype=torch.int64,device=dev); gathered=torch.empty((64,2*k+1),dtype=torch.int64,device=dev)
print("search host      ms", T(lambda: ix.search(q,k,1e3)))
print("search_packed    ms", T(lambda: ix.search_packed(q,k,1e3,packed.data_ptr())))
print("all_gather       ms", T(lambda: dist.all_gather_into_tensor(gathered,packed)))
print("merge device     ms", T(lambda: capi.topk_merge_packed_device(0,gathered.data_ptr(),1,64,k)))
print("alloc 2 tensors  ms", T(lambda: (torch.empty((64,2*k+1),dtype=torch.int64,device=dev), torch.empty((64,2*k+1),dtype=torch.int64,device=dev))))
from pixelbox_amd.sharded import ShardedIndex
sh=ShardedIndex(256,n,0,1,0,group=dist.group.WORLD); sh.index=ix
print("sharded.search   ms", T(lambda: sh.search(q,k,1e3)))
dist.destroy_process_group()
