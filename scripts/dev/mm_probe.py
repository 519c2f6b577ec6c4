import torch, time
for (M,K,N) in [(300000,431,200),(300000,200,200),(1000000,287,128),(1000000,543,256)]:
    x = torch.randn(M,K,device='cuda'); w = torch.randn(K,N,device='cuda'); y = torch.empty(M,N,device='cuda')
    for _ in range(3): torch.mm(x,w,out=y)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): torch.mm(x,w,out=y)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/10
    print(M,K,N, f'{t*1e6:.0f} us  {2*M*K*N/t/1e12:.1f} TFLOP/s  {(M*K+M*N)*4/t/1e12:.2f} TB/s')
