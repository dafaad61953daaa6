import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import gemm_bench as gb
from gp_amd import lib
M=131072
NOX=512
gb.bench(M,1024,1024,0,0,0,name="tall FF K=1024 xcd")
gb.bench(M,1024,1024,0,0,NOX,name="tall FF K=1024 plain")
gb.bench(M,1024,1024,0,0,lib.GEMM_KHI_N,name="V (KHI_N) xcd")
gb.bench(M,1024,1024,0,0,lib.GEMM_KHI_N|NOX,name="V (KHI_N) plain")
gb.bench(M,1024,1024,0,1,lib.GEMM_KLO_N,name="B (KLO_N) xcd")
gb.bench(M,1024,1024,0,1,lib.GEMM_KLO_N|NOX,name="B (KLO_N) plain")
gb.bench(M,2048,2048,0,0,0,name="tall m=2048 xcd")
gb.bench(M,2048,2048,0,0,NOX,name="tall m=2048 plain")
