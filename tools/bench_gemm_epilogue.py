import sys, os, torch
sys.path.insert(0, os.getcwd())
from image2text_amd import ops
from tools.bench_gemm import timeit
dev = torch.device('cuda:0'); BF16=torch.bfloat16
for (M,N,K) in [(16384,768,768),(16384,768,3072),(16384,3072,768),(16384,2304,768),(66560,512,512),(66560,2048,512)]:
    x = torch.randn(M,K,device=dev).to(BF16); w=(torch.randn(N,K,device=dev)*0.05).to(BF16)
    bias=torch.randn(N,device=dev); res=torch.randn(M,N,device=dev)
    yb=torch.empty(M,N,device=dev,dtype=BF16); yf=torch.empty(M,N,device=dev); pre=torch.empty(M,N,device=dev,dtype=BF16)
    fl=2.0*M*N*K
    t0=timeit(lambda: ops.gemm(x,w,yb,M,N,K))
    t1=timeit(lambda: ops.gemm(x,w,yb,M,N,K,bias=bias))
    t2=timeit(lambda: ops.gemm(x,w,yf,M,N,K,bias=bias,residual=res))
    t3=timeit(lambda: ops.gemm(x,w,yb,M,N,K,bias=bias,act=1,aux_out=pre))
    t4=timeit(lambda: ops.gemm(x,w,yf,M,N,K,bias=bias,residual=res,drop=(1,12345,429496729,1/0.9)))
    print(f'{M}x{N}x{K}: plain {fl/t0/1e12:6.1f}  bias {fl/t1/1e12:6.1f}  bias+res f32 {fl/t2/1e12:6.1f}  gelu+pre {fl/t3/1e12:6.1f}  res+dropout {fl/t4/1e12:6.1f} TF')
