#!/usr/bin/env python3
"""TPS kernel: time against the number of control points, with and without the sampling part."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from coupe.dvsg_amd import _lib
from warp_bench import timeit
dev=torch.device('cuda:0'); s=torch.cuda.current_stream().cuda_stream
B,H,W=16,720,1280
g=torch.Generator(device=dev).manual_seed(0)
U=torch.rand((B,H,W,3),generator=g,device=dev); out=torch.empty_like(U)
lin=torch.linspace(-1,1,5,device=dev)
coord=torch.stack(torch.meshgrid(lin,lin,indexing='xy'),-1).reshape(1,25,2).repeat(B,1,1).contiguous()
vec=0.05*torch.randn((B,25,2),generator=g,device=dev)
T=torch.empty((B,2,28),device=dev)
_lib.call("dvsg_tps_solve_f32",coord.data_ptr(),vec.data_ptr(),1,B,25,T.data_ptr(),s)
xs=torch.empty((B*H*W,),device=dev); ys=torch.empty_like(xs)
for P in (25,9,3):
    c=coord[:,:P].contiguous(); Tp=T[:,:,:P+3].contiguous()
    t_full=timeit(lambda:_lib.call("dvsg_tps_warp_f32",U.data_ptr(),c.data_ptr(),Tp.data_ptr(),B,H,W,3,P,H,W,out.data_ptr(),0,0,s))
    t_grid=timeit(lambda:_lib.call("dvsg_tps_warp_f32",0,c.data_ptr(),Tp.data_ptr(),B,H,W,3,P,H,W,0,xs.data_ptr(),ys.data_ptr(),s))
    print("P=%d full %.1f us  grid-only %.1f us"%(P,t_full[0],t_grid[0]))
