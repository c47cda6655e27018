import sys, os
sys.path[:0]=[os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'e4t-diffusion_amd'), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests')]
import torch
from e4t import ops
from emu_backend import EmuBackend
import kernel_checks as kc
hip=ops.HipBackend(); emu=EmuBackend(); dev=torch.device('cuda:0')
cases=[(1000, 200, 328, 256, 1), (2048, 256, 64, 256, 1), (700, 320, 1280, 256, 2), (700,320,1280,256,1), (512,256,128,256,1)]
for i,(M,N,K,tile,sk) in enumerate(cases):
    g=kc.gen(10+i,dev)
    a,b=kc.rnd(g,M,K,dev=dev),kc.rnd(g,N,K,scale=K**-0.5,dev=dev)
    try:
        y=hip.gemm(a,b,tile=tile,splitk=sk); torch.cuda.synchronize()
        print(M,N,K,tile,sk,'rel',kc.rel(y,emu.gemm(a,b)),flush=True)
    except Exception as e:
        print(M,N,K,tile,sk,'EXC',e,flush=True); break
