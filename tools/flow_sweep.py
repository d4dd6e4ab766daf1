#!/usr/bin/env python3
"""tf_warp bandwidth against the coherence of the flow field (zero, constant, N(0, 4 px) box-smoothed)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from coupe.dvsg_amd import _lib
from warp_bench import timeit
dev=torch.device('cuda:0'); s=torch.cuda.current_stream().cuda_stream
B,H,W=64,720,1280
g=torch.Generator(device=dev).manual_seed(0)
U=torch.rand((B,H,W,3),generator=g,device=dev); out=torch.empty_like(U)
def run(name, flow):
    med,mn=timeit(lambda:_lib.call("dvsg_flow_warp_f32",U.data_ptr(),flow.data_ptr(),B,H,W,3,out.data_ptr(),s))
    print("%-28s %8.1f us %7.1f GB/s"%(name,med,32.0*B*H*W/med/1e3))
run("zero flow", torch.zeros((B,H,W,2),device=dev))
f=torch.zeros((B,H,W,2),device=dev); f[...,0]=3.3; f[...,1]=1.7
run("constant (3.3,1.7)", f)
for box in (61,31,15,7):
    fl=4.0*box*torch.randn((B,2,H,W),generator=g,device=dev)
    fl=torch.nn.functional.avg_pool2d(fl,box,stride=1,padding=box//2).permute(0,2,3,1).contiguous()
    run("N(0,4) box %d"%box, fl)
